"""hipGraph replay of a whole detection train step (ocrs_models_amd/graph.py) must be the eager step: same loss sequence, same parameters
after K steps -- fp32 parity mode and bf16 throughput mode, on changing input batches (the graph reads its static input buffers)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_train_step_equals_eager(dev, dtype):
    import ocrs_models_amd as oa

    B, S, K = 2, 128, 4
    r = np.random.RandomState(5)
    xs = [torch.from_numpy(r.uniform(-0.5, 0.5, (B, 1, S, S)).astype(np.float32)).to(dev) for _ in range(K)]
    ts = [torch.from_numpy((r.uniform(0, 1, (B, 1, S, S)) > 0.9).astype(np.float32)).to(dev) for _ in range(K)]

    def make():
        torch.manual_seed(1234)
        m = oa.DetectionModel(act_dtype=dtype).to(dev)
        m.train()
        return m

    # eager
    m1 = make()
    o1 = oa.optim.Adam(m1.parameters())
    eager = []
    for x, t in zip(xs, ts):
        loss = oa.balanced_cross_entropy_loss(m1(x), t)
        o1.zero_grad()
        loss.backward()
        o1.step()
        eager.append(float(loss))
    # graphed: the warm-up / capture steps run on a throw-away copy of the state so that both runs start from the same point
    m2 = make()
    o2 = oa.optim.Adam(m2.parameters(), capturable=True)
    sd = {k: v.clone() for k, v in m2.state_dict().items()}
    step = oa.graph.GraphedTrainStep(m2, o2, oa.balanced_cross_entropy_loss, xs[0], ts[0])
    # after construction the host mirror of the step count equals the device-side count (the capture itself executes no Adam kernel)
    torch.cuda.synchronize()
    for ds in o2._dev_step.values():
        assert all(st["step"] == int(ds.item()) for st in o2.state.values()), ([st["step"] for st in o2.state.values()][:3], float(ds.item()))
    m2.load_state_dict(sd)  # (in place: the recorded pointers stay valid)
    for st in o2.state.values():
        st["exp_avg"].zero_()
        st["exp_avg_sq"].zero_()
        st["step"] = 0
    for ds in o2._dev_step.values():
        ds.zero_()
    graphed = [float(step(x, t)) for x, t in zip(xs, ts)]
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-5  # same kernels, same order: bit-reproducible reductions
    assert np.allclose(graphed, eager, rtol=tol, atol=0), (graphed, eager)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-7), k
    assert all(st["step"] == K for st in o2.state.values())
