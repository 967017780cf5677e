"""artiboost_amd -- MI355X-native (gfx950) implementation of ArtiBoost's online-synthesis + pose-training hot path.

Host side mirrors the reference's `anakin` plugin surface; all device work goes through the C ABI in
include/artiboost_hip.h (libartiboost_hip.so, hand-written HIP)."""
__version__ = "0.1.0"
