"""GPU test of the pair / two-tile BPTT experiments (tools/bwd_decompositions/libdeepj_bwd_exp.so; build.sh first)
against the production per-tile kernel of libdeepj_hip.so.  Not part of tests/: the kernels are not part of the product.

    sh tools/bwd_decompositions/build.sh && python -m pytest tools/bwd_decompositions/test_bwd_pair_dual.py -q
"""
import ctypes as C
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
_P = C.c_void_p


def _lib():
    from music_generator_amd import _lib as L
    return L, L.load()


def _exp():
    lib = C.CDLL(os.path.join(HERE, "libdeepj_bwd_exp.so"))
    sig = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int32, _P, _P]
    for n in ("dj_lstm_bwd_pair", "dj_lstm_bwd_dual"):
        getattr(lib, n).restype, getattr(lib, n).argtypes = C.c_int32, sig
    lib.dj_bwd_exp_scratch_bytes.restype = C.c_int64
    lib.dj_bwd_exp_faults.restype, lib.dj_bwd_exp_faults.argtypes = C.c_int32, [_P, _P]
    return lib


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("entry", ["dj_lstm_bwd_pair", "dj_lstm_bwd_dual"])
@pytest.mark.parametrize("tiles,Ls", [(8, 4), (9, 1), (17, 33), (2, 2), (64, 6), (301, 3)])
def test_lstm_bwd_pair_tile_counts(gpu_device, tiles, Ls, entry):
    """dj_lstm_bwd_pair / dj_lstm_bwd_dual (two tiles per workgroup pair, interleaved) against dj_lstm_bwd on the same
    (random) stash / cell states / upstream gradient: pair groups that are full, partly filled and absent; a single
    step (no exchange at all); odd tile counts (the dual form hands its last tile to the per-tile kernel); more tiles
    than one launch holds (301 > 256: two launches); column-tile-major dZ.  The two kernels sum dz U^T in a different k order, so dz agrees
    to bf16 rounding of nearly equal fp32 sums, not bit for bit."""
    L, lib = _lib()
    exp = _exp()
    H, R = 256, tiles * Ls * 32
    g = torch.Generator().manual_seed(100 + tiles)
    U = torch.randn(H, 4 * H, generator=g) * (1.0 / H ** 0.5)
    upb = torch.empty(H * 4 * H * 2, dtype=torch.uint8, device=gpu_device)
    L.check(lib.dj_lstm_pack(1, H, L.ptr(U.to(gpu_device)), None, L.ptr(upb), _st()), "pack")
    Z = torch.randint(0, 256, (R * 4 * H,), generator=g, dtype=torch.uint8).to(gpu_device)       # 8-bit gate codes
    Cc = (torch.randn(R * H, generator=g) * 0.7).to(torch.bfloat16).to(gpu_device)
    dH = (torch.randn(R, H, generator=g) * 0.1).to(torch.bfloat16).to(gpu_device)
    cts = R * 256
    outs = []
    cl = torch.zeros(exp.dj_bwd_exp_scratch_bytes(), dtype=torch.uint8, device=gpu_device)
    for pair in (False, True):
        dZ = torch.zeros(4 * cts, dtype=torch.bfloat16, device=gpu_device)
        db = torch.zeros(4 * H, dtype=torch.float32, device=gpu_device)
        if pair:
            L.check(getattr(exp, entry)(1, H, tiles, Ls, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ), cts,
                                        L.ptr(db), 0, L.ptr(cl), _st()), entry)
        else:
            L.check(lib.dj_lstm_bwd(1, H, tiles, Ls, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ), cts, L.ptr(db),
                                    0, _st()), "bwd")
        torch.cuda.synchronize()
        outs.append((dZ.float().cpu(), db.cpu()))
    assert exp.dj_bwd_exp_faults(L.ptr(cl), _st()) == 0
    (dz0, db0), (dz1, db1) = outs
    assert torch.isfinite(dz1).all()
    scale = float(dz0.abs().max())
    assert scale > 0
    assert float((dz1 - dz0).abs().max()) <= 1e-2 * scale
    assert float((dz1 - dz0).abs().mean()) <= 2e-4 * scale
    torch.testing.assert_close(db1, db0, rtol=2e-3, atol=2e-3 * float(db0.abs().max()))
