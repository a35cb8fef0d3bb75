"""Op-level parity of the bidirectional GRU recurrence (reference: nn.GRU at ocrs_models/models.py:245, 264-266): the per-step kernels
(csrc/rec_gru.hip) and the persistent one-launch-per-layer form (csrc/rec_gru_seq.hip, exact-fp32 and split-bf16 x3 arithmetic) against
torch.nn.GRU and an explicit float64 recurrence with autograd (which also yields the gradients w.r.t. the pre-activations gi / gh that the
C ABI exposes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, G3 = 256, 768


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _weights(seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / np.sqrt(H)
    whh = (torch.rand(2, G3, H, generator=g) * 2 - 1) * k * scale
    bhh = (torch.rand(2, G3, generator=g) * 2 - 1) * k
    return whh, bhh


def _ref(gi, whh, bhh, dout=None):
    """float64 recurrence, both directions.  gi [T][N][2*768].  Returns out [T][N][512] (+ dgi, dgh when dout is given)."""
    T, N = gi.shape[:2]
    gi = gi.double().clone().requires_grad_(dout is not None)
    whh, bhh = whh.double(), bhh.double()
    gh_bias = bhh.view(1, 1, 2, G3).expand(T, N, 2, G3).clone().requires_grad_(dout is not None)  # per-(t, b) copy: its grad is dL/dgh
    outs = [[None] * T, [None] * T]
    for d in (0, 1):
        h = torch.zeros(N, H, dtype=torch.float64)
        for s in range(T):
            t = s if d == 0 else T - 1 - s
            gh = h @ whh[d].T + gh_bias[t, :, d]
            gx = gi[t, :, d * G3:(d + 1) * G3]
            r = torch.sigmoid(gx[:, :H] + gh[:, :H])
            z = torch.sigmoid(gx[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gx[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            outs[d][t] = h
    out = torch.stack([torch.cat([outs[0][t], outs[1][t]], 1) for t in range(T)], 0)
    if dout is None:
        return out.detach()
    out.backward(dout.double())
    return out.detach(), gi.grad, gh_bias.grad.reshape(T, N, 2 * G3)


def _hip_fwd(L, mode, gi, whh, bhh, T, N, dev, train=True):
    from ocrs_models_amd._lib import ptr

    out = torch.empty(T, N, 512, device=dev)
    saved = torch.empty(T, N, 2, 4, 256, device=dev) if train else None
    if mode == "step":
        nfl = 8 * 48 * 64 * 8
        pk = torch.empty(2 * nfl, device=dev)
        for d in (0, 1):
            L.pack_frags(whh.data_ptr() + 4 * d * G3 * H, 0, 256, 768, 256, 0, 1, 256, pk.data_ptr() + 4 * d * nfl, 0)
        L.gru_layer_fwd(ptr(gi), ptr(pk), ptr(bhh), ptr(out), ptr(saved), T, N)
    else:
        sync = torch.empty(L.gru_seq_sync_words(N), dtype=torch.int32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        xws = torch.empty(L.gru_seq_ws_floats(N), device=dev)
        L.gru_seq_fwd(ptr(gi), ptr(whh), ptr(bhh), ptr(out), ptr(saved), T, N, ptr(sync), ptr(err), ptr(xws), 1 if mode == "seq_exact" else 0)
        L.gru_seq_status(ptr(err))
        assert int(err.item()) == 0
    return out, saved


def _hip_bwd(L, mode, dout, saved, out, whh, T, N, dev):
    from ocrs_models_amd._lib import ptr

    dgi = torch.empty(T, N, 2 * G3, device=dev)
    dgh = torch.empty(T, N, 2 * G3, device=dev)
    if mode == "step":
        nfl = 24 * 16 * 64 * 8
        pk = torch.empty(2 * nfl, device=dev)
        for d in (0, 1):
            L.pack_frags(whh.data_ptr() + 4 * d * G3 * H, 0, 768, 256, 768, 0, 256, 1, pk.data_ptr() + 4 * d * nfl, 0)
        dhz = torch.empty(2, 2, N, 256, device=dev)
        L.gru_layer_bwd(ptr(dout), ptr(saved), ptr(out), ptr(pk), ptr(dgi), ptr(dgh), ptr(dhz), T, N)
    else:
        sync = torch.empty(L.gru_seq_sync_words(N), dtype=torch.int32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        xws = torch.empty(L.gru_seq_ws_floats(N), device=dev)
        # the persistent launch also accumulates the bias gradients (column sums of dgi / dgh) into caller-owned, pre-filled buffers
        dbih, dbhh = torch.full((2 * G3,), 0.5, device=dev), torch.full((2 * G3,), -0.25, device=dev)
        L.gru_seq_bwd(ptr(dout), ptr(saved), ptr(out), ptr(whh), ptr(dgi), ptr(dgh), T, N, ptr(sync), ptr(err), ptr(xws), 1 if mode == "seq_exact" else 0,
                      ptr(dbih), ptr(dbhh))
        L.gru_seq_status(ptr(err))
        assert int(err.item()) == 0
        si, sh = dgi.double().sum((0, 1)), dgh.double().sum((0, 1))
        assert float(((dbih.double() - 0.5) - si).norm() / (si.norm() + 1e-30)) < 2e-5
        assert float(((dbhh.double() + 0.25) - sh).norm() / (sh.norm() + 1e-30)) < 2e-5
    return dgi, dgh


# tolerances: exact fp32 MFMA forms: fp32 rounding of a K = 256 / 768 dot product chained over T steps (measured ~1e-6);
# split-bf16 x3: 2^-17-class products (measured ~1e-5 forward, ~3e-5 backward) -- 4x the measured values
TOLS = {"step": (5e-6, 2e-5), "seq_exact": (5e-6, 2e-5), "seq_x3": (1e-4, 3e-4)}


@pytest.mark.parametrize("mode", ["step", "seq_exact", "seq_x3"])
@pytest.mark.parametrize("T,N", [(1, 1), (7, 5), (23, 40), (101, 64)])
def test_gru_recurrence_forward_backward_vs_float64(dev, mode, T, N):
    from ocrs_models_amd._lib import lib

    L = lib()
    if mode != "step" and not L.gru_seq_supported(N):
        pytest.skip("persistent GRU launch not resident on this device")
    whh, bhh = _weights(T * 100 + N)
    g = torch.Generator().manual_seed(T + N)
    gi = torch.randn(T, N, 2 * G3, generator=g)
    dout = torch.randn(T, N, 512, generator=g)
    out_r, dgi_r, dgh_r = _ref(gi, whh, bhh, dout)
    gi_d, whh_d, bhh_d, dout_d = gi.to(dev), whh.to(dev).contiguous(), bhh.to(dev).contiguous(), dout.to(dev)
    out, saved = _hip_fwd(L, mode, gi_d, whh_d, bhh_d, T, N, dev)
    dgi, dgh = _hip_bwd(L, mode, dout_d, saved, out, whh_d, T, N, dev)
    tf, tb = TOLS[mode]
    assert rel(out, out_r) < tf, rel(out, out_r)
    assert rel(dgi, dgi_r) < tb, rel(dgi, dgi_r)
    assert rel(dgh, dgh_r) < tb, rel(dgh, dgh_r)


def test_gru_forward_matches_torch_nn_gru(dev):
    """the whole layer (projection in torch, recurrence on the GPU) against torch.nn.GRU (CPU, float64): bidirectional, PyTorch gate order"""
    from ocrs_models_amd._lib import lib

    L = lib()
    T, N, I = 19, 33, 128
    torch.manual_seed(5)
    gru = torch.nn.GRU(I, H, bidirectional=True).double()
    x = torch.randn(T, N, I, dtype=torch.float64)
    ref, _ = gru(x)
    w_ih = torch.cat([gru.weight_ih_l0, gru.weight_ih_l0_reverse], 0)
    b_ih = torch.cat([gru.bias_ih_l0, gru.bias_ih_l0_reverse], 0)
    gi = (x @ w_ih.T + b_ih).float().to(dev).contiguous()
    whh = torch.stack([gru.weight_hh_l0, gru.weight_hh_l0_reverse], 0).float().to(dev).contiguous()
    bhh = torch.cat([gru.bias_hh_l0, gru.bias_hh_l0_reverse], 0).float().to(dev).contiguous()
    modes = ["step"] + (["seq_exact", "seq_x3"] if L.gru_seq_supported(N) else [])
    for mode in modes:
        out, _ = _hip_fwd(L, mode, gi, whh, bhh, T, N, dev, train=False)
        assert rel(out, ref) < (1e-4 if mode == "seq_x3" else 5e-6), (mode, rel(out, ref))


def test_gru_persistent_same_xcd_path_equals_agent_scope_path(dev, monkeypatch):
    """the verified same-XCD exchange (plain stores + L2 arrival counter) and the placement-independent agent-scope exchange move the same
    values: outputs are bit-identical"""
    from ocrs_models_amd._lib import lib

    L = lib()
    T, N = 101, 256
    if not L.gru_seq_supported(N):
        pytest.skip("persistent GRU launch not resident on this device")
    whh, bhh = _weights(11)
    g = torch.Generator().manual_seed(11)
    gi = torch.randn(T, N, 2 * G3, generator=g).to(dev)
    dout = torch.randn(T, N, 512, generator=g).to(dev)
    whh, bhh = whh.to(dev).contiguous(), bhh.to(dev).contiguous()
    for mode in ("seq_exact", "seq_x3"):
        res = {}
        for fast in ("0", "1"):
            monkeypatch.setenv("OCRS_GRU_SEQ_FAST", fast)
            out, saved = _hip_fwd(L, mode, gi, whh, bhh, T, N, dev)
            dgi, dgh = _hip_bwd(L, mode, dout, saved, out, whh, T, N, dev)
            res[fast] = (out.clone(), dgi.clone(), dgh.clone())
        for a, b in zip(res["0"], res["1"]):
            assert torch.equal(a, b)


def test_gru_persistent_is_deterministic_and_repeatable(dev):
    """two launches on the same inputs are bit-identical (fixed summation order, no atomics on data), also right after each other on the same
    buffers (arrival counters are re-zeroed by every call)"""
    from ocrs_models_amd._lib import lib

    L = lib()
    T, N = 101, 256
    if not L.gru_seq_supported(N):
        pytest.skip("persistent GRU launch not resident on this device")
    whh, bhh = _weights(9)
    g = torch.Generator().manual_seed(9)
    gi = torch.randn(T, N, 2 * G3, generator=g).to(dev)
    dout = torch.randn(T, N, 512, generator=g).to(dev)
    whh, bhh = whh.to(dev).contiguous(), bhh.to(dev).contiguous()
    for mode in ("seq_exact", "seq_x3"):
        res = []
        for _ in range(3):
            out, saved = _hip_fwd(L, mode, gi, whh, bhh, T, N, dev)
            dgi, dgh = _hip_bwd(L, mode, dout, saved, out, whh, T, N, dev)
            res.append((out.clone(), dgi.clone(), dgh.clone()))
        for r in res[1:]:
            for a, b in zip(res[0], r):
                assert torch.equal(a, b)
