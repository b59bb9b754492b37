from .data_preprocessor import DataPreprocessor
from .feature_preprocessor import FeaturePreprocessor
from .chunk_preprocessor import ChunkPreprocessor

__all__ = ["DataPreprocessor", "FeaturePreprocessor", "ChunkPreprocessor"]
