"""MI355X-native Swin detection hot path.

The package directory is also reachable as ``swin-transformer-object-detection_amd``
(a symlink): Python package names cannot contain hyphens.
"""
__version__ = "0.1.0"
