"""Single-kernel parity tests: each C-ABI kernel entry point against the CPU oracle /
a plain torch-CPU fp32 reference of the same op.  Run on the GPU box: -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import deepj_oracle as O

pytestmark = pytest.mark.gpu

DT = {"f32": 0, "bf16": 1}


def _lib():
    from music_generator_amd import _lib as L
    return L, L.load()


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _op(t, dtype):
    return t.to(torch.bfloat16 if dtype == "bf16" else torch.float32).contiguous()


def _tol(dtype):
    return (2e-2, 2e-2) if dtype == "bf16" else (2e-4, 2e-5)


def to_rows(x):
    """[S, L, D] -> kernel row order [(tile, step, s32), D] with zero-padded tiles."""
    S, L, D = x.shape
    tiles = (S + 31) // 32
    xp = torch.zeros(tiles * 32, L, D, dtype=x.dtype)
    xp[:S] = x
    return xp.reshape(tiles, 32, L, D).permute(0, 2, 1, 3).reshape(tiles * L * 32, D).contiguous(), tiles


def from_rows(r, S, L):
    D = r.shape[1]
    tiles = r.shape[0] // (32 * L)
    return r.reshape(tiles, L, 32, D).permute(0, 2, 1, 3).reshape(tiles * 32, L, D)[:S]


def to_frag(rows):
    """row-major [R, Ccols] -> fragment-tiled flat (include/deepj_hip.h, dj_gemm_nt c_mode 2)."""
    R, Cc = rows.shape
    x = rows.reshape(R // 32, 32, Cc // 32, 32)             # rb, s, cb, c
    s = torch.arange(32)
    lane_hi = (s >> 2) & 1
    reg = (s & 3) + 4 * (s >> 3)
    out = torch.zeros(R // 32, Cc // 32, 64, 16, dtype=rows.dtype)
    for si in range(32):
        out[:, :, lane_hi[si] * 32:(lane_hi[si] + 1) * 32, reg[si]] = x[:, si, :, :]
    return out.reshape(-1).contiguous()


def from_frag(flat, R, Cc):
    f = flat.reshape(R // 32, Cc // 32, 64, 16)
    s = torch.arange(32)
    lane_hi = (s >> 2) & 1
    reg = (s & 3) + 4 * (s >> 3)
    x = torch.zeros(R // 32, 32, Cc // 32, 32, dtype=flat.dtype)
    for si in range(32):
        x[:, si, :, :] = f[:, :, lane_hi[si] * 32:(lane_hi[si] + 1) * 32, reg[si]]
    return x.reshape(R, Cc)


def stash_buffer(R, H, dtype, device):
    """Empty gate stash for R rows of a layer with H units (include/deepj_hip.h 'Gate stash')."""
    L, lib = _lib()
    n = lib.dj_lstm_stash_bytes(DT[dtype], H, R)
    assert n == R * 4 * H * (4 if dtype == "f32" else 1)
    return torch.zeros(R * 4 * H, dtype=torch.float32 if dtype == "f32" else torch.uint8, device=device)


def check_stash(G, R, H, S, Ls, dtype, sigm, Zref):
    """The stash against the reference pre-activations Zref [S, Ls, 4H]: fp32 holds z itself; bf16 holds the
    activated gates as 8-bit codes -- decoded values within half a code step (+ bf16 noise of z), and codes 0 / 255
    exactly where hard_sigmoid is saturated."""
    rt, at = _tol(dtype)
    g = from_rows(from_frag(G.float().cpu(), R, 4 * H), S, Ls)
    if dtype == "f32":
        torch.testing.assert_close(g, Zref, rtol=rt, atol=at * 10)
        return
    ract = torch.sigmoid if sigm else O.hard_sigmoid
    for k in (0, 1, 3):
        code, z = g[..., k * H:(k + 1) * H], Zref[..., k * H:(k + 1) * H]
        val = ((code - 0.5) / 254).clamp(0, 1)
        assert float((val - ract(z)).abs().max()) <= 0.5 / 254 + 0.2 * 3e-2 + 1e-6
        if not sigm:        # saturation codes, away from the +-2.5 boundary by more than the bf16 noise of z
            assert bool((code[z < -2.6] == 0).all()) and bool((code[z > 2.6] == 255).all())
            inside = z.abs() < 2.4
            assert bool(((code[inside] >= 1) & (code[inside] <= 254)).all())
    code, z = g[..., 2 * H:3 * H], Zref[..., 2 * H:3 * H]
    assert float(((code - 128) / 127 - torch.tanh(z)).abs().max()) <= 0.5 / 127 + 3e-2 + 1e-6


def test_dropout_mask_matches_oracle(gpu_device):
    L, lib = _lib()
    for seed, site, p, rows, cols in [(7, 4, 0.5, 1000, 64), (123456789012, 17, 0.2, 333, 259), (0, 1, 0.2, 64, 3)]:
        m = torch.empty(rows, cols, dtype=torch.float32, device=gpu_device)
        L.check(lib.dj_dropout_mask(C.c_uint64(seed), site, p, rows, cols, L.ptr(m), _st()), "mask")
        ref = O.keep_mask(seed, site, rows, cols, p).astype(np.float32) * np.float32(1.0 / (1.0 - np.float32(p)))
        np.testing.assert_array_equal(m.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (96, 94, 96), (768, 1024, 96), (160, 259, 512), (1024, 512, 264),
                                   (4100, 264, 512), (300, 200, 1024), (2304, 640, 40), (70000, 256, 96)])
def test_gemm_nt(gpu_device, dtype, M, N, K):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    Bt = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    Ad, Bd = _op(A, dtype).to(gpu_device), _op(Bt, dtype).to(gpu_device)
    ldc = ((N + 7) // 8) * 8
    Cd = torch.zeros(M, ldc, dtype=Ad.dtype, device=gpu_device)
    L.check(lib.dj_gemm_nt(DT[dtype], M, N, K, L.ptr(Ad), K, L.ptr(Bd), K, L.ptr(Cd), ldc, 0,
                           L.ptr(bias.to(gpu_device)), _st()), "gemm_nt")
    ref = Ad.float().cpu() @ Bd.float().cpu().T + bias
    rt, at = _tol(dtype)
    torch.testing.assert_close(Cd.float().cpu()[:, :N], ref, rtol=rt, atol=at * K ** 0.5)
    if ldc > N:
        assert float(Cd[:, N:].abs().max()) == 0.0
    if dtype == "bf16" and K % 256 == 0:
        # A column-tile-major [K/256][M][256] (the BPTT sweep's dZ): same product
        At = Ad.reshape(M, K // 256, 256).permute(1, 0, 2).contiguous()
        Ct = torch.zeros_like(Cd)
        L.check(lib.dj_gemm_nt_tiled_a(DT[dtype], M, N, K, L.ptr(At), M * 256, L.ptr(Bd), K, L.ptr(Ct), ldc, 0,
                                       L.ptr(bias.to(gpu_device)), _st()), "gemm_nt tiled A")
        assert torch.equal(Ct, Cd)
    if dtype == "bf16":
        # the epilogue writes rows with 16-byte stores when C, bias and ldc allow it (store_block_rows16: the two halves of a
        # wave trade pieces); a C that is only 8-byte aligned takes the 8-byte path: same values, nothing outside [M, N]
        buf = torch.zeros(M * ldc + 12, dtype=Ad.dtype, device=gpu_device)
        L.check(lib.dj_gemm_nt(DT[dtype], M, N, K, L.ptr(Ad), K, L.ptr(Bd), K, C.c_void_p(buf.data_ptr() + 8), ldc, 0,
                               L.ptr(bias.to(gpu_device)), _st()), "gemm_nt, C 8-byte aligned")
        assert torch.equal(buf[4:4 + M * ldc].view(M, ldc), Cd)
        assert float(buf[:4].abs().max()) == 0.0 and float(buf[4 + M * ldc:].abs().max()) == 0.0
    if dtype == "bf16":
        # c_mode 3: C += A Bt^T + bias in place (the per-step recurrent product of the scaled model's forward sweep)
        C0 = _op(torch.randn(M, ldc, generator=g), dtype)
        Ca = C0.clone().to(gpu_device)
        L.check(lib.dj_gemm_nt(DT[dtype], M, N, K, L.ptr(Ad), K, L.ptr(Bd), K, L.ptr(Ca), ldc, 3,
                               L.ptr(bias.to(gpu_device)), _st()), "gemm_nt acc")
        torch.testing.assert_close(Ca.float().cpu()[:, :N], C0.float()[:, :N] + ref, rtol=rt, atol=at * K ** 0.5)
        if ldc > N:
            assert torch.equal(Ca.cpu()[:, N:], C0[:, N:])           # padding columns untouched
    else:
        assert lib.dj_gemm_nt(DT[dtype], M, N, K, L.ptr(Ad), K, L.ptr(Bd), K, L.ptr(Cd), ldc, 3, None, _st()) >= 1000
    if M % 32 == 0 and N % 32 == 0:
        Cf = torch.zeros(M * N, dtype=Ad.dtype, device=gpu_device)
        L.check(lib.dj_gemm_nt(DT[dtype], M, N, K, L.ptr(Ad), K, L.ptr(Bd), K, L.ptr(Cf), N, 2,
                               L.ptr(bias.to(gpu_device)), _st()), "gemm_nt frag")
        torch.testing.assert_close(from_frag(Cf.float().cpu(), M, N), ref, rtol=rt, atol=at * K ** 0.5)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("M,Ka,kv,N,shift,steps", [(1024, 96, 94, 512, 0, 0), (2048, 264, 259, 512, 0, 0),
                                                   (1536, 128, 128, 512, 32, 6), (4096, 256, 256, 1024, 32, 8),
                                                   # small outputs (bf16: one accumulator tile per wave): the conv
                                                   # kernel gradient's shape, row counts that leave waves idle, narrower operands
                                                   (98304, 80, 72, 64, 0, 0), (960, 80, 72, 64, 0, 0),
                                                   (64, 96, 96, 64, 0, 0), (4992, 40, 33, 16, 0, 0)])
def test_gemm_tn(gpu_device, dtype, M, Ka, kv, N, shift, steps):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + Ka)
    A = torch.randn(M, Ka, generator=g)
    B = torch.randn(M, N, generator=g) * 0.1
    Ad, Bd = _op(A, dtype).to(gpu_device), _op(B, dtype).to(gpu_device)
    Cd = torch.full((kv, N), 0.5, dtype=torch.float32, device=gpu_device)
    L.check(lib.dj_gemm_tn(DT[dtype], M, Ka, kv, N, L.ptr(Ad), Ka, L.ptr(Bd), N, L.ptr(Cd), N, shift, steps, _st()),
            "gemm_tn")
    Af = Ad.float().cpu()
    if shift:
        As = torch.zeros_like(Af)
        As[32:] = Af[:-32]
        blk = (torch.arange(M) // 32) % steps
        As[blk == 0] = 0
        Af = As
    ref = 0.5 + Af.T[:kv] @ Bd.float().cpu()
    rt, at = _tol(dtype)
    torch.testing.assert_close(Cd.cpu(), ref, rtol=rt, atol=at * M ** 0.5)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tiles,steps,DP,D,H,N", [(4, 6, 96, 94, 256, 1024), (3, 5, 264, 259, 128, 512),
                                                  (8, 4, 128, 128, 128, 512), (2, 3, 256, 256, 256, 1024)])
def test_lstm_wgrad_fused(gpu_device, dtype, tiles, steps, DP, D, H, N):
    """dW = X^T dZ and dU = Hprev^T dZ in one launch vs a torch fp32 reference."""
    L, lib = _lib()
    M = tiles * steps * 32
    g = torch.Generator().manual_seed(M + DP)
    X = torch.randn(M, DP, generator=g)
    X[:, D:] = 0
    Hs = torch.randn(M, H, generator=g)
    dZ = torch.randn(M, N, generator=g) * 0.1
    Xd, Hd, Zd = [_op(t, dtype).to(gpu_device) for t in (X, Hs, dZ)]
    dW = torch.full((D, N), 0.25, dtype=torch.float32, device=gpu_device)
    dU = torch.full((H, N), -0.5, dtype=torch.float32, device=gpu_device)
    zeros = torch.zeros(64, dtype=torch.float32, device=gpu_device)
    L.check(lib.dj_lstm_wgrad(DT[dtype], M, steps, L.ptr(Xd), DP, D, L.ptr(Hd), H, L.ptr(Zd), N, 0, L.ptr(dW), L.ptr(dU),
                              L.ptr(zeros), _st()), "wgrad")
    if dtype == "bf16":
        # the same product from a column-tile-major dZ ([N/256][M][256]; what the bf16 BPTT sweep writes)
        Zt = Zd.reshape(M, N // 256, 256).permute(1, 0, 2).contiguous()
        dW2, dU2 = torch.full_like(dW, 0.25), torch.full_like(dU, -0.5)
        L.check(lib.dj_lstm_wgrad(DT[dtype], M, steps, L.ptr(Xd), DP, D, L.ptr(Hd), H, L.ptr(Zt), N, M * 256, L.ptr(dW2),
                                  L.ptr(dU2), L.ptr(zeros), _st()), "wgrad tiled")
        torch.testing.assert_close(dW2, dW, rtol=1e-5, atol=1e-4)       # same products, atomics in another order
        torch.testing.assert_close(dU2, dU, rtol=1e-5, atol=1e-4)
    else:
        assert lib.dj_lstm_wgrad(DT[dtype], M, steps, L.ptr(Xd), DP, D, L.ptr(Hd), H, L.ptr(Zd), N, M * 256, L.ptr(dW),
                                 L.ptr(dU), L.ptr(zeros), _st()) >= 1000
    Hp = torch.zeros(M, H)
    Hp[32:] = Hd.float().cpu()[:-32]
    Hp[((torch.arange(M) // 32) % steps) == 0] = 0
    rt, at = _tol(dtype)
    torch.testing.assert_close(dW.cpu(), 0.25 + Xd.float().cpu()[:, :D].T @ Zd.float().cpu(), rtol=rt,
                               atol=at * M ** 0.5)
    torch.testing.assert_close(dU.cpu(), -0.5 + Hp.T @ Zd.float().cpu(), rtol=rt, atol=at * M ** 0.5)


def _lstm_setup(S, Ls, D, H, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(S, Ls, D, generator=g) * 0.5
    W = torch.randn(D, 4 * H, generator=g) * (1.0 / D ** 0.5)
    U = torch.randn(H, 4 * H, generator=g) * (1.0 / H ** 0.5)
    b = torch.randn(4 * H, generator=g) * 0.1
    return x, W, U, b


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("H,S,Ls,sigm", [(256, 96, 7, 0), (128, 40, 12, 0), (256, 32, 5, 1), (128, 64, 3, 1)])
def test_lstm_fwd_bwd(gpu_device, dtype, H, S, Ls, sigm):
    L, lib = _lib()
    D = 24
    x, W, U, b = _lstm_setup(S, Ls, D, H, H + S)
    cfg = O.OracleConfig(recurrent_activation="sigmoid" if sigm else "hard_sigmoid")
    # the kernel consumes operand-typed x*W+b and U; mirror that rounding in the reference
    rnd = (lambda t: t.to(torch.bfloat16).float()) if dtype == "bf16" else (lambda t: t)
    zx = rnd(x @ W + b)
    Ur = rnd(U)

    zx_ref = zx.clone().requires_grad_(True)
    Uref = Ur.clone().requires_grad_(True)
    ract = O.hard_sigmoid if not sigm else torch.sigmoid
    h = torch.zeros(S, H)
    c = torch.zeros(S, H)
    hs, cs, zs = [], [], []
    for t in range(Ls):
        z = zx_ref[:, t] + h @ Uref
        i, f, gg, o = ract(z[:, :H]), ract(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), ract(z[:, 3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        hs.append(h); cs.append(c); zs.append(z)
    Href, Cref, Zref = torch.stack(hs, 1), torch.stack(cs, 1), torch.stack(zs, 1)
    dH = torch.randn(S, Ls, H, generator=torch.Generator().manual_seed(5)) * 0.1
    dHr = rnd(dH)
    (Href * dHr).sum().backward()

    zrows, tiles = to_rows(zx)
    R = zrows.shape[0]
    Zd = _op(to_frag(zrows), dtype).to(gpu_device)
    Ud = U.to(gpu_device)
    esz = 2 if dtype == "bf16" else 4
    upf = torch.empty(H * 4 * H * esz, dtype=torch.uint8, device=gpu_device)
    upb = torch.empty(H * 4 * H * esz, dtype=torch.uint8, device=gpu_device)
    L.check(lib.dj_lstm_pack(DT[dtype], H, L.ptr(Ud), L.ptr(upf), L.ptr(upb), _st()), "pack")
    Hd = torch.zeros(tiles * Ls * 32, H, dtype=Zd.dtype, device=gpu_device)
    Cd = torch.zeros(R * H, dtype=Zd.dtype, device=gpu_device)
    Gd = stash_buffer(R, H, dtype, gpu_device)
    if dtype == "bf16":       # the 8-bit stash must not overwrite the bf16 projections
        assert lib.dj_lstm_fwd(DT[dtype], H, tiles, Ls, L.ptr(Zd), L.ptr(Zd), L.ptr(upf), L.ptr(Hd), L.ptr(Cd), sigm,
                               _st()) >= 1000
    L.check(lib.dj_lstm_fwd(DT[dtype], H, tiles, Ls, L.ptr(Zd), L.ptr(Gd), L.ptr(upf), L.ptr(Hd), L.ptr(Cd), sigm, _st()),
            "fwd")
    rt, at = _tol(dtype)
    torch.testing.assert_close(from_rows(Hd.float().cpu(), S, Ls), Href.detach(), rtol=rt, atol=at * 5)
    torch.testing.assert_close(from_rows(from_frag(Cd.float().cpu(), R, H), S, Ls), Cref.detach(), rtol=rt,
                               atol=at * 5)
    check_stash(Gd, R, H, S, Ls, dtype, sigm, Zref.detach())

    dHd = _op(to_rows(dH)[0], dtype).to(gpu_device)
    db = torch.zeros(4 * H, dtype=torch.float32, device=gpu_device)
    dZd = torch.zeros(R, 4 * H, dtype=Zd.dtype, device=gpu_device)
    L.check(lib.dj_lstm_bwd(DT[dtype], H, tiles, Ls, L.ptr(Gd), L.ptr(upb), L.ptr(Cd), L.ptr(dHd), L.ptr(dZd), 0,
                            L.ptr(db), sigm, _st()), "bwd")
    # column-tile-major dZ ([4H/256][rows][256], with slack between the tiles): the same numbers elsewhere
    cts = R * 256 + 512
    dZt = torch.zeros((4 * H // 256) * cts, dtype=Zd.dtype, device=gpu_device)
    db2 = torch.zeros_like(db)
    L.check(lib.dj_lstm_bwd(DT[dtype], H, tiles, Ls, L.ptr(Gd), L.ptr(upb), L.ptr(Cd), L.ptr(dHd), L.ptr(dZt), cts,
                            L.ptr(db2), sigm, _st()), "bwd tiled")
    back = dZt.reshape(4 * H // 256, cts)[:, :R * 256].reshape(4 * H // 256, R, 256).permute(1, 0, 2).reshape(R, 4 * H)
    assert torch.equal(back, dZd)
    assert lib.dj_lstm_bwd(DT[dtype], H, tiles, Ls, L.ptr(Gd), L.ptr(upb), L.ptr(Cd), L.ptr(dHd), L.ptr(dZt), R * 256 - 1,
                           L.ptr(db2), sigm, _st()) >= 1000
    dz = from_rows(dZd.float().cpu(), S, Ls)
    torch.testing.assert_close(dz, zx_ref.grad, rtol=rt * 2, atol=at * 5)
    torch.testing.assert_close(db.cpu(), zx_ref.grad.sum(dim=(0, 1)), rtol=rt * 2, atol=at * 20)
    # padded sequences of the last tile must produce exactly zero dz
    if S % 32:
        full = dZd.float().cpu().reshape(tiles, Ls, 32, 4 * H)
        assert float(full[-1, :, S % 32:, :].abs().max()) == 0.0



def test_nadam_matches_oracle(gpu_device):
    from music_generator_amd import engine
    n = 10007
    rs = np.random.RandomState(3)
    p = rs.randn(n).astype(np.float32)
    st = O.NadamState()
    opt = engine.Nadam(n, gpu_device)
    pd = torch.from_numpy(p.copy()).to(gpu_device)
    pref = p.copy()
    for it in range(5):
        g = (rs.randn(n) * 0.1).astype(np.float32)
        pref = O.nadam_step(pref, g, st)
        opt.step(pd, torch.from_numpy(g).to(gpu_device))
    np.testing.assert_allclose(pd.cpu().numpy(), pref, rtol=2e-5, atol=2e-6)
    # and the oracle itself against torch.optim.NAdam (same recurrence)
    tp = torch.nn.Parameter(torch.from_numpy(p.copy()))
    topt = torch.optim.NAdam([tp], lr=0.002, betas=(0.9, 0.999), eps=1e-8, momentum_decay=0.004)
    rs = np.random.RandomState(3)
    rs.randn(n)
    for it in range(5):
        tp.grad = torch.from_numpy((rs.randn(n) * 0.1).astype(np.float32))
        topt.step()
    np.testing.assert_allclose(tp.detach().numpy(), pref, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("H,S,Ls,D", [(256, 64, 5, 94), (128, 40, 6, 259), (256, 32, 3, 256), (128, 96, 4, 128),
                                      (128, 40, 6, 90),
                                      # bf16 H = 256 takes the weight-stationary cluster kernel at every tile count
                                      # (2, 1, 64, 256 tiles here: partly idle clusters, full grid)
                                      (256, 2048, 6, 256), (256, 8192, 3, 94),
                                      # two cluster launches: 256 tiles + 75 tiles on a grid of 128 wave slots
                                      (256, 331 * 32, 2, 94), (256, 150, 9, 94)])
def test_lstm_fwd_fused_input_projection(gpu_device, djenv, dtype, H, S, Ls, D):
    """z = x W + h U + b inside the recurrent kernel (dj_lstm_fwd_fused) vs the restated cell."""
    _fwd_fused_case(gpu_device, dtype, H, S, Ls, D, 0, djenv)


@pytest.mark.parametrize("H,S,Ls,D", [(256, 2048, 4, 256), (256, 96, 4, 94), (128, 64, 5, 128)])
def test_lstm_fwd_fused_recurrent_sigmoid(gpu_device, djenv, H, S, Ls, D):
    """The same kernels with recurrent_activation='sigmoid' (template switch; Keras' default is hard_sigmoid), bf16:
    cluster kernel (64 tiles), per-tile H = 256, register-stationary H = 128."""
    _fwd_fused_case(gpu_device, "bf16", H, S, Ls, D, 1, djenv)


def _fwd_fused_case(gpu_device, dtype, H, S, Ls, D, sigm, djenv=None):
    L, lib = _lib()
    ract = torch.sigmoid if sigm else O.hard_sigmoid
    DP = (D + 7) // 8 * 8
    x, W, U, b = _lstm_setup(S, Ls, D, H, H + D)
    rnd = (lambda t: t.to(torch.bfloat16).float()) if dtype == "bf16" else (lambda t: t)
    xr, Wr, Ur = rnd(x), rnd(W), rnd(U)
    h = torch.zeros(S, H); c = torch.zeros(S, H)
    hs, cs, zs = [], [], []
    for t in range(Ls):
        z = xr[:, t] @ Wr + b + h @ Ur
        i, f, gg, o = ract(z[:, :H]), ract(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), ract(z[:, 3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        hs.append(h); cs.append(c); zs.append(z)
    Href, Cref, Zref = torch.stack(hs, 1), torch.stack(cs, 1), torch.stack(zs, 1)
    xp = torch.zeros(S, Ls, DP); xp[:, :, :D] = x
    xrows, tiles = to_rows(xp)
    R = xrows.shape[0]
    Xd = _op(xrows, dtype).to(gpu_device)
    esz = 2 if dtype == "bf16" else 4
    wpack = torch.zeros(4 * H * (DP + 128) * esz, dtype=torch.uint8, device=gpu_device)
    upf = torch.empty(H * 4 * H * esz, dtype=torch.uint8, device=gpu_device)
    L.check(lib.dj_lstm_pack_w(DT[dtype], H, L.ptr(W.contiguous().to(gpu_device)), D, L.ptr(wpack), _st()), "packw")
    L.check(lib.dj_lstm_pack(DT[dtype], H, L.ptr(U.to(gpu_device)), L.ptr(upf), None, _st()), "pack")
    Gd = stash_buffer(R, H, dtype, gpu_device)
    Hd = torch.zeros(R, H, dtype=Xd.dtype, device=gpu_device)
    Cd = torch.zeros(R * H, dtype=Xd.dtype, device=gpu_device)
    # exchange state of the cluster kernel (bf16 H = 256 from 64 tiles): caller-owned, zero-initialised once
    cl = torch.zeros(lib.dj_lstm_cluster_scratch_bytes(), dtype=torch.uint8, device=gpu_device)
    rc = lib.dj_lstm_fwd_fused(DT[dtype], H, tiles, Ls, L.ptr(Xd), DP, D, L.ptr(wpack), L.ptr(b.to(gpu_device)),
                               L.ptr(Gd), L.ptr(upf), L.ptr(Hd), L.ptr(Cd), sigm, L.ptr(cl), _st())
    L.check(rc, "fwd_fused")
    rt, at = _tol(dtype)
    torch.testing.assert_close(from_rows(Hd.float().cpu(), S, Ls), Href, rtol=rt, atol=at * 5)
    torch.testing.assert_close(from_rows(from_frag(Cd.float().cpu(), R, H), S, Ls), Cref, rtol=rt, atol=at * 5)
    check_stash(Gd, R, H, S, Ls, dtype, sigm, Zref)
    assert lib.dj_lstm_cluster_faults(L.ptr(cl), _st()) == 0      # no wait expired, every cluster sat on one XCD
    if dtype == "bf16" and H == 256:
        # the same sweep without the scratch takes the per-tile kernel: same h to the last bit is not promised
        # (different summation order), same numbers to bf16 tolerance is
        Hd2 = torch.zeros_like(Hd)
        L.check(lib.dj_lstm_fwd_fused(DT[dtype], H, tiles, Ls, L.ptr(Xd), DP, D, L.ptr(wpack), L.ptr(b.to(gpu_device)),
                                      None, L.ptr(upf), L.ptr(Hd2), None, sigm, None, _st()), "fwd_fused per-tile")
        torch.testing.assert_close(Hd2.float().cpu(), Hd.float().cpu(), rtol=rt, atol=at * 5)
        if djenv is not None:
            # the cluster sweep's two exchange protocols -- h slices that announce themselves by a tag in a spare exponent
            # bit (default) and the per-step counter (DJ_KF_COUNTED_EXCHANGE) -- are the same sums in the same order:
            # h, c and the gate stash must agree to the last bit
            djenv.set("DEEPJ_TAGGED_EXCHANGE", "0")
            Hd3 = torch.zeros_like(Hd); Cd3 = torch.zeros_like(Cd); Gd3 = stash_buffer(R, H, dtype, gpu_device)
            L.check(lib.dj_lstm_fwd_fused(DT[dtype], H, tiles, Ls, L.ptr(Xd), DP, D, L.ptr(wpack), L.ptr(b.to(gpu_device)),
                                          L.ptr(Gd3), L.ptr(upf), L.ptr(Hd3), L.ptr(Cd3), sigm, L.ptr(cl), _st()), "counted")
            djenv.unset("DEEPJ_TAGGED_EXCHANGE")
            assert torch.equal(Hd3, Hd) and torch.equal(Cd3, Cd) and torch.equal(Gd3, Gd)
            assert lib.dj_lstm_cluster_faults(L.ptr(cl), _st()) == 0


@pytest.mark.parametrize("D", [128, 90])
def test_lstm_bwd_fused_input_gradient(gpu_device, D):
    """dj_lstm_bwd_dx (bf16, H = 128): the dX the BPTT kernel produces from its dz tile equals dZ W^T computed
    from the dZ it wrote; wider inputs are refused."""
    L, lib = _lib()
    H, tiles, Ls = 128, 3, 5
    R, DP = tiles * Ls * 32, (D + 7) // 8 * 8
    g = torch.Generator().manual_seed(D)
    bf = lambda t: t.to(torch.bfloat16).to(gpu_device)
    Z = torch.randint(0, 256, (R * 4 * H,), generator=g, dtype=torch.uint8).to(gpu_device)   # any 8-bit gate stash
    Cc = bf(torch.randn(R * H, generator=g) * 0.5)
    dH = bf(torch.randn(R, H, generator=g) * 0.1)
    U = (torch.randn(H, 4 * H, generator=g) * 0.05).to(gpu_device)
    W = (torch.randn(D, 4 * H, generator=g) * 0.05).to(gpu_device)
    upf = torch.empty(H * 4 * H * 2, dtype=torch.uint8, device=gpu_device); upb = torch.empty_like(upf)
    L.check(lib.dj_lstm_pack(1, H, L.ptr(U), L.ptr(upf), L.ptr(upb), _st()), "pack")
    wt = torch.zeros(((D + 31) // 32) * 32 * 4 * H * 2, dtype=torch.uint8, device=gpu_device)
    L.check(lib.dj_lstm_pack_wt(1, H, L.ptr(W), D, L.ptr(wt), _st()), "pack_wt")
    dZ = torch.zeros(R, 4 * H, dtype=torch.bfloat16, device=gpu_device)
    dX = torch.full((R, DP), 7.0, dtype=torch.bfloat16, device=gpu_device)
    db = torch.zeros(4 * H, device=gpu_device)
    L.check(lib.dj_lstm_bwd_dx(1, H, tiles, Ls, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ), 0, L.ptr(db), 0,
                               L.ptr(wt), D, L.ptr(dX), DP, _st()), "bwd_dx")
    assert float(dZ.float().abs().max()) > 0
    ref = dZ.float().cpu() @ W.to(torch.bfloat16).float().cpu().T
    got = dX.float().cpu()
    torch.testing.assert_close(got[:, :D], ref, rtol=2e-2, atol=2e-2 * float(ref.abs().max()))
    if DP > D:
        assert float(got[:, D:].abs().max()) == 0.0
    # the same sweep without the fused gradient writes the same dZ
    dZ2 = torch.zeros_like(dZ); db2 = torch.zeros_like(db)
    L.check(lib.dj_lstm_bwd(1, H, tiles, Ls, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ2), 0, L.ptr(db2), 0,
                            _st()), "bwd")
    assert torch.equal(dZ, dZ2)


def test_lstm_bwd_remainder_input_gradient(gpu_device):
    """dj_lstm_bwd_dx with D = 259 (note layer 0: 256 time-axis features + 3 chosen columns): the kernel
    produces only the last 32-column block (4 columns stored), K split over its waves and folded through LDS
    one step later; columns 0..255 are left to the GEMM."""
    L, lib = _lib()
    H, tiles, Ls, D, DP = 128, 3, 6, 259, 264
    R = tiles * Ls * 32
    g = torch.Generator().manual_seed(5)
    bf = lambda t: t.to(torch.bfloat16).to(gpu_device)
    Z = torch.randint(0, 256, (R * 4 * H,), generator=g, dtype=torch.uint8).to(gpu_device)
    Cc = bf(torch.randn(R * H, generator=g) * 0.5)
    dH = bf(torch.randn(R, H, generator=g) * 0.1)
    U = (torch.randn(H, 4 * H, generator=g) * 0.05).to(gpu_device)
    W = (torch.randn(D, 4 * H, generator=g) * 0.05).to(gpu_device)
    upf = torch.empty(H * 4 * H * 2, dtype=torch.uint8, device=gpu_device); upb = torch.empty_like(upf)
    L.check(lib.dj_lstm_pack(1, H, L.ptr(U), L.ptr(upf), L.ptr(upb), _st()), "pack")
    wt = torch.zeros(((D + 31) // 32) * 32 * 4 * H * 2, dtype=torch.uint8, device=gpu_device)
    L.check(lib.dj_lstm_pack_wt(1, H, L.ptr(W), D, L.ptr(wt), _st()), "pack_wt")
    dZ = torch.zeros(R, 4 * H, dtype=torch.bfloat16, device=gpu_device)
    dX = torch.full((R, DP), 7.0, dtype=torch.bfloat16, device=gpu_device)
    db = torch.zeros(4 * H, device=gpu_device)
    L.check(lib.dj_lstm_bwd_dx(1, H, tiles, Ls, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ), 0, L.ptr(db), 0,
                               L.ptr(wt), D, L.ptr(dX), DP, _st()), "bwd_dx")
    ref = dZ.float().cpu() @ W.to(torch.bfloat16).float().cpu().T          # [R, 259]
    got = dX.float().cpu()
    torch.testing.assert_close(got[:, 256:259], ref[:, 256:259], rtol=2e-2, atol=2e-2 * float(ref.abs().max()))
    assert float(got[:, 259].abs().max()) == 0.0                           # 4th stored column: zero weight row
    assert bool((got[:, :256] == 7.0).all()) and bool((got[:, 260:] == 7.0).all())


@pytest.mark.parametrize("H,S,Ls", [(256, 96, 9), (128, 64, 12), (256, 64, 128), (128, 64, 128)])
def test_gate_stash_quantiser_bounds_dz(gpu_device, H, S, Ls):
    """The 8-bit activated-gate stash of the bf16 kernels (include/deepj_hip.h 'Gate stash': |error| <= 1/508 for
    i, f, o and 1/254 for g, codes 0 / 255 = saturated hard_sigmoid) tested as a QUANTISER: the same forward is run by
    the bf16 kernels (8-bit stash) and by the fp32 kernels (z stash, exact activations in BPTT) on identical operand
    values; BPTT of both on the same upstream gradient; the difference in dz and in the bias gradient is bounded at
    ~2x what the encoding explains, far below the whole-step gradient tolerance.  Pre-activations are scaled so that
    a good share of the gates saturate (codes 0 and 255 both present).  The 128-step cases are the reference's window
    (constants.py:67): the stash error of every step feeds the recurrent gradient of all earlier steps."""
    L, lib = _lib()
    D, sigm = 24, 0
    x, W, U, b = _lstm_setup(S, Ls, D, H, 3 * H + S)
    bf = lambda t: t.to(torch.bfloat16).float()
    zx = bf((x @ W + b) * 2.5)                 # |z| spreads past the +-2.5 knees of hard_sigmoid
    Ur = bf(U)
    dH = bf(torch.randn(S, Ls, H, generator=torch.Generator().manual_seed(5)) * 0.1)
    zrows, tiles = to_rows(zx)
    R = zrows.shape[0]
    res = {}
    for dtype in ("f32", "bf16"):
        esz = 2 if dtype == "bf16" else 4
        Zd = _op(to_frag(zrows), dtype).to(gpu_device)
        upf = torch.empty(H * 4 * H * esz, dtype=torch.uint8, device=gpu_device)
        upb = torch.empty_like(upf)
        L.check(lib.dj_lstm_pack(DT[dtype], H, L.ptr(Ur.to(gpu_device)), L.ptr(upf), L.ptr(upb), _st()), "pack")
        Hd = torch.zeros(R, H, dtype=Zd.dtype, device=gpu_device)
        Cd = torch.zeros(R * H, dtype=Zd.dtype, device=gpu_device)
        Gd = stash_buffer(R, H, dtype, gpu_device)
        L.check(lib.dj_lstm_fwd(DT[dtype], H, tiles, Ls, L.ptr(Zd), L.ptr(Gd), L.ptr(upf), L.ptr(Hd), L.ptr(Cd), sigm,
                                _st()), "fwd")
        dHd = _op(to_rows(dH)[0], dtype).to(gpu_device)
        db = torch.zeros(4 * H, dtype=torch.float32, device=gpu_device)
        dZd = torch.zeros(R, 4 * H, dtype=Zd.dtype, device=gpu_device)
        L.check(lib.dj_lstm_bwd(DT[dtype], H, tiles, Ls, L.ptr(Gd), L.ptr(upb), L.ptr(Cd), L.ptr(dHd), L.ptr(dZd), 0,
                                L.ptr(db), sigm, _st()), "bwd")
        res[dtype] = (from_rows(dZd.float().cpu(), S, Ls), db.cpu(), from_rows(from_frag(Gd.float().cpu(), R, 4 * H), S, Ls),
                      from_rows(Hd.float().cpu(), S, Ls))
    dz32, db32, z32, h32 = res["f32"]
    dz16, db16, code, h16 = res["bf16"]
    # the forward passes agree to bf16 rounding of h (the recurrence feeds it back)
    assert float((h16 - h32).abs().max()) < 2e-2
    # saturation is exercised on both sides, and the reserved codes mark exactly the saturated gates of the fp32 run
    gates = torch.cat([z32[..., :2 * H], z32[..., 3 * H:]], -1)
    codes = torch.cat([code[..., :2 * H], code[..., 3 * H:]], -1)
    assert float((codes == 0).float().mean()) > 0.02 and float((codes == 255).float().mean()) > 0.02
    clear = (gates.abs() - 2.5).abs() > 0.1          # away from the knee by more than the bf16 noise of z
    assert bool(((codes == 0) == (gates < -2.5))[clear].all()) and bool(((codes == 255) == (gates > 2.5))[clear].all())
    # a saturated gate has derivative 0: dz of that gate is exactly zero in both runs
    dzg16 = torch.cat([dz16[..., :2 * H], dz16[..., 3 * H:]], -1)
    assert float(dzg16[(codes == 0) | (codes == 255)].abs().max()) == 0.0
    # dz of a gate whose pre-activation sits AT a knee of hard_sigmoid can differ by the whole value between the two
    # runs (the bf16 forward's z noise flips the 0.2 / 0 derivative): that is forward rounding, not the stash -- the
    # element-wise bound is taken over the gates that are clear of the knees (g has none); the rms and the bias
    # gradient are taken over everything
    ok = torch.ones_like(dz32, dtype=torch.bool)
    ok[..., :2 * H] = clear[..., :2 * H]
    ok[..., 3 * H:] = clear[..., 2 * H:]
    assert float(ok.float().mean()) > 0.9
    scale = float(dz32.abs().max())
    err = float((dz16 - dz32)[ok].abs().max()) / scale
    err_all = float((dz16 - dz32).abs().max()) / scale
    rms = float((dz16 - dz32).pow(2).mean().sqrt()) / float(dz32.pow(2).mean().sqrt())
    dberr = float((db16 - db32).abs().max()) / float(db32.abs().max())
    print("stash quantiser H=%d: max|ddz|/max|dz| %.2e clear of the knees (%.2e with them), rms ratio %.2e, dbias %.2e"
          % (H, err, err_all, rms, dberr))
    assert err < 1.5e-2 and rms < 1.5e-2 and dberr < 2.5e-2, (err, err_all, rms, dberr)


@pytest.mark.parametrize("S,Ls,sigm", [(96, 9, 0), (40, 4, 1)])
def test_lstm_bwd256_split_vs_plain_kernel(gpu_device, djenv, S, Ls, sigm):
    """bf16 H = 256 BPTT: the split-gate-math sweep (lstm_bwd256_kernel, the default since round 5: dh-independent factors
    computed inside the product loop and kept as fp16 pairs, k-major dz tile, dH by LDS-DMA) against the round-4 kernel it
    replaces (DJ_KF_BWD_PLAIN / DEEPJ_BWD_SPLIT=0), on ONE forward stash, upstream gradient and packed U^T: the same dZ
    (row-major and column-tile-major) and bias gradient up to the rounding of the fp16 factors and of bf16 dz -- and the
    fallback kernel stays exercised now that it is no longer the default.  S = 40 leaves 24 padded sequences in the last
    tile: exactly zero dz there in both."""
    L, lib = _lib()
    H, D = 256, 24
    x, W, U, b = _lstm_setup(S, Ls, D, H, H + S + 1)
    zx = (x @ W + b).to(torch.bfloat16).float()
    zrows, tiles = to_rows(zx)
    R = zrows.shape[0]
    Zd = _op(to_frag(zrows), "bf16").to(gpu_device)
    Ud = U.to(gpu_device)
    upf = torch.empty(H * 4 * H * 2, dtype=torch.uint8, device=gpu_device)
    upb = torch.empty_like(upf)
    L.check(lib.dj_lstm_pack(1, H, L.ptr(Ud), L.ptr(upf), L.ptr(upb), _st()), "pack")
    Hd = torch.zeros(tiles * Ls * 32, H, dtype=Zd.dtype, device=gpu_device)
    Cd = torch.zeros(R * H, dtype=Zd.dtype, device=gpu_device)
    Gd = stash_buffer(R, H, "bf16", gpu_device)
    L.check(lib.dj_lstm_fwd(1, H, tiles, Ls, L.ptr(Zd), L.ptr(Gd), L.ptr(upf), L.ptr(Hd), L.ptr(Cd), sigm, _st()), "fwd")
    dH = torch.randn(S, Ls, H, generator=torch.Generator().manual_seed(9)) * 0.1
    dHd = _op(to_rows(dH)[0], "bf16").to(gpu_device)
    cts = R * 256 + 256

    def run(split):
        djenv.set("DEEPJ_BWD_SPLIT", "1" if split else "0")
        dZ = torch.zeros(R, 4 * H, dtype=Zd.dtype, device=gpu_device)
        db = torch.zeros(4 * H, dtype=torch.float32, device=gpu_device)
        L.check(lib.dj_lstm_bwd(1, H, tiles, Ls, L.ptr(Gd), L.ptr(upb), L.ptr(Cd), L.ptr(dHd), L.ptr(dZ), 0, L.ptr(db), sigm,
                                _st()), "bwd")
        dZt = torch.zeros(4 * cts, dtype=Zd.dtype, device=gpu_device)
        db2 = torch.zeros_like(db)
        L.check(lib.dj_lstm_bwd(1, H, tiles, Ls, L.ptr(Gd), L.ptr(upb), L.ptr(Cd), L.ptr(dHd), L.ptr(dZt), cts, L.ptr(db2),
                                sigm, _st()), "bwd tiled")
        torch.cuda.synchronize()
        back = dZt.reshape(4, cts)[:, :R * 256].reshape(4, R, 256).permute(1, 0, 2).reshape(R, 4 * H)
        assert torch.equal(back, dZ)                  # both dZ layouts of one kernel: the same numbers
        return dZ.float().cpu(), db.cpu()

    dz_s, db_s = run(True)
    dz_p, db_p = run(False)
    scale = float(dz_p.abs().max())
    assert scale > 1e-3
    err = float((dz_s - dz_p).abs().max()) / scale
    dberr = float((db_s - db_p).abs().max()) / float(db_p.abs().max())
    print("split vs plain BPTT sweep: max |d dz| / max |dz| = %.2e, bias gradient %.2e" % (err, dberr))
    assert err < 1.5e-2 and dberr < 5e-3, (err, dberr)
    if S % 32:
        for dz in (dz_s, dz_p):
            assert float(dz.reshape(tiles, Ls, 32, 4 * H)[-1, :, S % 32:, :].abs().max()) == 0.0
