"""medplib_amd — MI355X-native (gfx950) hot path for MedPLIB: HIP kernels behind a C ABI, thin Python host mirror."""
__version__ = "0.1.0"
