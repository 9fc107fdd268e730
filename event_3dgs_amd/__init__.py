"""MI355X-native hot path of Event-3DGS: differentiable Gaussian rasteriser + event loss.

Product code.  Never imports anything from oracle/ (test infrastructure)."""
__version__ = "0.1.0"
