from .base import MultipleShooting

__all__ = ["MultipleShooting"]
