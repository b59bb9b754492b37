from .mp_pool import mp_pool, mp_worker

__all__ = ["mp_pool", "mp_worker"]
