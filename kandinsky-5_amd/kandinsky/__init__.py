"""kandinsky — MI355X-native drop-in for the hot path of ai-forever/Kandinsky-5 (T2V Lite).

Same import surface as the reference package (`from kandinsky import get_T2V_pipeline`,
kandinsky/__init__.py:1); the DiT sampling loop and the VAE decode run in libk5.so (HIP, gfx950).
"""
from .utils import get_T2V_pipeline  # noqa: F401
