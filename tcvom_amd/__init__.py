"""tcvom_amd — MI355X-native (gfx950) implementation of the TCVOM per-frame-window hot path
(GCA base matting network + Temporal Attention Module).  See DESIGN.md."""
__version__ = '0.1.0'
