from .lpips import LPIPS

__all__ = ["LPIPS"]
