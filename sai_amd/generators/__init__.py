from .chunk_generator import ChunkGenerator
from .data_generator import DataGenerator
from .window_generator import WindowGenerator

__all__ = ["DataGenerator", "ChunkGenerator", "WindowGenerator"]
