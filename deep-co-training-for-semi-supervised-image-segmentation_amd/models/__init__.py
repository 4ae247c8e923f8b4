from .segmentators import Segmentator  # noqa: F401
