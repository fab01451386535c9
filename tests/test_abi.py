"""CPU: the C-ABI library builds/loads and exports every symbol include/medplib_hip.h declares (no compute calls)."""
import ctypes
import os

from medplib_amd import _lib


def test_header_parses_and_library_exports_every_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 30
    if not os.path.exists(_lib.LIB_PATH):
        from medplib_amd import build
        build.build(verbose=False)
    dll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(dll, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_loader_signatures_and_error_plumbing():
    L = _lib.lib()
    assert L.raw("mp_version")() >= 100
    assert L.raw("mp_arch")() == b"gfx950"
    # argument validation happens before any launch, so it is safe without a GPU
    rc = L.raw("mp_gemm_bf16_nt")(None, 64, None, 64, None, 64, None, None, 0, 4, 4, 65, 0, 0, 1.0, None, None)
    assert rc == -1 and "multiple of 64" in L.last_error()
    rc = L.raw("mp_attention_fwd_bf16")(None, 0, 0, None, 0, 0, None, 0, 0, None, 0, 0, None, None, None, 0, 0, 1, 1, 1, 1, 48,
                                        0, 1.0, 0, None, None)
    assert rc == -1 and "head_dim" in L.last_error()


def test_no_cpu_fallback():
    """The product op layer refuses CPU tensors instead of silently computing somewhere else."""
    import pytest
    import torch
    from medplib_amd import ops
    with pytest.raises(ValueError):
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64), 1e-5)
