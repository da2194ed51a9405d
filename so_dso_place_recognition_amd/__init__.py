"""MI355X-native place-recognition hot path (generate_signatures + match_signatures of
IRVLab/so_dso_place_recognition).  Compute lives in libpr_amd.so (HIP, gfx950) behind the C ABI of
include/place_recognition.h; this package is the Python host-side mirror of the reference's interfaces."""
__all__ = ["api", "matcher", "synth"]
