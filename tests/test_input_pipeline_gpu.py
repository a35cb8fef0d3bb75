"""SURVEY 8(f) rows 1 and 3 on the GPU: the device-side input pipeline (transform_image, collate_samples, antialiased resize) against the
oracle, and the two validation loops (``test()``) against the oracle's eval-mode forward + loss.

Tolerances: transform / collate are byte -> fp32 maps with one IEEE divide and one subtract: BIT-EXACT.  The resize is a pair of short fp32
dot products (<= ~2*scale+1 terms, weights restated from ATen): 2e-6 absolute on values in [-0.5, 0.5].  Validation loss: the fp32 model
tolerances of test_det_model_gpu.py / test_rec_gpu.py."""
import numpy as np
import pytest
import torch

from tests.golden_util import REC_CASE, load_npz, rec_samples

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 256, 64 * 400 + 3, 1 << 20])
def test_transform_image_bit_exact(dev, n):
    from ocrs_models_amd import input_pipeline as ip
    from oracle import text as otext

    g = torch.Generator().manual_seed(n)
    img = torch.randint(0, 256, (1, 1, n), generator=g, dtype=torch.uint8) if n != 256 else torch.arange(256, dtype=torch.uint8).reshape(1, 1, 256)
    want = otext.transform_image(img)
    got = ip.transform_image(img.to(dev))
    assert got.dtype == torch.float32 and got.shape == img.shape
    assert torch.equal(got.cpu(), want)
    got16 = ip.transform_image(img.to(dev), dtype=torch.bfloat16)
    assert torch.equal(got16.cpu(), want.bfloat16())


def test_transform_image_rejects_host_and_float_input(dev):
    from ocrs_models_amd import input_pipeline as ip

    with pytest.raises(RuntimeError):
        ip.transform_image(torch.zeros(1, 4, 4, dtype=torch.uint8))
    with pytest.raises(RuntimeError):
        ip.transform_image(torch.zeros(1, 4, 4, device=dev))


def _u8_samples(seed, widths, lengths, h=64):
    g = torch.Generator().manual_seed(seed)
    out = []
    for w, L in zip(widths, lengths):
        out.append({"image": torch.randint(0, 256, (1, h, w), generator=g, dtype=torch.uint8),
                    "text_seq": torch.randint(1, 97, (L,), generator=g, dtype=torch.int32)})
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_collate_samples_matches_oracle(dev, dtype):
    """uint8 crops: fused transform + bucketed padding == oracle collate of the transformed samples, incl. the round_up quirk
    (max width 256 -> bucket 512), a width-1 crop, and an infeasible sample (L > w // 4) that must be dropped."""
    from ocrs_models_amd import input_pipeline as ip
    from oracle import text as otext

    widths = [37, 118, 200, 256, 1, 10, 255, 13, 64, 101, 3, 77, 250, 199, 8, 31]
    lengths = [5, 20, 40, 64, 1, 2, 33, 9, 16, 25, 1, 12, 62, 7, 5, 7]  # sample 7 (w=13 -> 3 steps, L=9) and 14 (w=8 -> 2, L=5) infeasible
    raw = _u8_samples(5, widths, lengths)
    want = otext.collate([{"image": otext.transform_image(s["image"]), "text_seq": s["text_seq"]} for s in raw])
    got = ip.collate_samples(raw, dev, dtype=dtype)
    assert want["image"].shape[0] < len(raw)  # something was dropped
    assert got["image"].is_cuda and got["image"].dtype == dtype and tuple(got["image"].shape) == tuple(want["image"].shape)
    assert torch.equal(got["image"].cpu(), want["image"].to(dtype))
    for k in ("text_seq", "text_len", "image_width"):
        assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k


def test_collate_samples_float_input_matches_golden_batch(dev):
    """fp32 samples (the reference's sample format) through the device collate == the batch captured from the imported reference."""
    from ocrs_models_amd import input_pipeline as ip
    from tests.golden_util import compare_to_golden

    G = load_npz("rec.npz")
    got = ip.collate_samples(rec_samples(REC_CASE), dev)
    assert tuple(got["image"].shape) == tuple(G["rec1/batch/image_shape"])
    # the golden holds norm / sum / samples of the image (float64 reductions whose last bit depends on the buffer's alignment) ...
    assert compare_to_golden(G, "rec1/batch/image", got["image"].cpu(), 0) < 1e-12
    # ... and the oracle collate, which the CPU suite pins to the same golden, must match bit for bit
    from oracle import text as otext

    assert torch.equal(got["image"].cpu(), otext.collate(rec_samples(REC_CASE))["image"])
    assert np.array_equal(got["text_seq"].numpy(), G["rec1/batch/text_seq"])
    assert np.array_equal(got["text_len"].numpy(), G["rec1/batch/text_len"])
    assert np.array_equal(got["image_width"].numpy(), G["rec1/batch/image_width"])


@pytest.mark.parametrize("h,w", [(37, 211), (120, 900), (64, 400), (20, 9), (200, 3000), (5, 40), (64, 1), (1, 64), (333, 2500)])
def test_resize_line_matches_oracle(dev, h, w):
    from ocrs_models_amd import input_pipeline as ip
    from oracle import input_pipe as oip

    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.rand(1, h, w, generator=g) - 0.5
    ow = oip.line_output_width(h, w)
    assert ip.line_output_width(h, w) == ow and 10 <= ow <= 800
    want = oip.resize_aa(x, [64, ow])
    got = ip.resize_line(x.to(dev))
    assert tuple(got.shape) == (1, 64, ow)
    assert (got.cpu() - want).abs().max().item() < 2e-6


def test_resize_batched_planes(dev):
    from ocrs_models_amd import input_pipeline as ip
    from oracle import input_pipe as oip

    x = torch.rand(3, 2, 50, 70, generator=torch.Generator().manual_seed(1)) - 0.5
    got = ip.resize(x.to(dev), [33, 41])
    assert (got.cpu() - oip.resize_aa(x, [33, 41])).abs().max().item() < 2e-6


def test_pipeline_end_to_end_feeds_the_recognition_model(dev):
    """uint8 line crops -> transform -> resize to 64 rows -> collate -> RecognitionModel: shapes and value range the model expects."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import input_pipeline as ip

    g = torch.Generator().manual_seed(2)
    samples = []
    for (h, w) in [(30, 200), (48, 600), (90, 350)]:
        u8 = torch.randint(0, 256, (1, h, w), generator=g, dtype=torch.uint8).to(dev)
        line = ip.resize_line(ip.transform_image(u8))
        assert line.shape[1] == 64 and -0.5 <= float(line.min()) and float(line.max()) <= 0.5
        samples.append({"image": line.cpu(), "text_seq": torch.randint(1, 97, (6,), generator=g, dtype=torch.int32)})
    batch = ip.collate_samples(samples, dev)
    m = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET).to(dev).eval()
    with torch.no_grad():
        lp = m(batch["image"])
    assert lp.shape == (batch["image"].shape[-1] // 4 + 1, 3, 97) and torch.isfinite(lp).all()


def _load(m, seed, specs):
    from oracle.params import make_state, state_dict_from

    P, Bf = make_state(specs, seed)
    m.load_state_dict(state_dict_from(P, Bf, specs))
    return m, P, Bf


def test_detection_validation_loop_matches_oracle(dev):
    """train_detection.test(): eval-mode forward + balanced BCE, mean over batches; metrics hook is called once per image."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import train_detection as td
    from oracle import detection as odet
    from oracle import losses as olosses
    from oracle.params import detection_specs

    m, P, Bf = _load(oa.DetectionModel(), 21, detection_specs())
    m = m.to(dev)
    g = torch.Generator().manual_seed(4)
    batches = []
    for b in (2, 1):
        batches.append({"image": torch.rand(b, 1, 64, 96, generator=g) - 0.5, "text_mask": (torch.rand(b, 1, 64, 96, generator=g) > 0.8).float(),
                        "path": ["x"] * b})
    want = 0.0
    with torch.no_grad():
        for bt in batches:
            pred = odet.forward(P, Bf, bt["image"], False)
            want += float(olosses.balanced_bce(pred, bt["text_mask"]))
    want /= len(batches)
    calls = []

    def metrics_fn(bp, bm):
        assert not bp.is_cuda and set(bp.unique().tolist()) <= {0.0, 1.0}
        calls.append(1)
        return {"precision": 0.5, "recall": float(len(calls))}

    loss, metrics = td.test(dev, batches, m, metrics_fn=metrics_fn)
    assert not m.training
    assert abs(loss - want) < 2e-4 * abs(want), (loss, want)
    assert len(calls) == 3 and metrics["precision"] == 0.5 and abs(metrics["recall"] - 2.0) < 1e-12
    loss2, metrics2 = td.test(dev, batches, m, metrics_fn=None)
    assert loss2 == loss and metrics2 == {}
    # default: the word-level metrics of the reference's loop (postprocess.py restated in ocrs_models_amd/postprocess.py), the mean over the
    # images of what mask_metrics gives for the binarised prediction and target of each
    from ocrs_models_amd.postprocess import mask_metrics
    loss3, metrics3 = td.test(dev, batches, m)
    assert loss3 == loss and set(metrics3) == {"precision", "recall", "merged_frac", "split_frac"}
    per_image = []
    with torch.inference_mode():
        for bt in batches:
            pr = m(bt["image"].to(dev))
            for i in range(pr.shape[0]):
                per_image.append(mask_metrics(td.binarize_mask(pr[i]).cpu(), td.binarize_mask(bt["text_mask"][i])))
    for k in metrics3:
        assert abs(metrics3[k] - sum(d[k] for d in per_image) / len(per_image)) < 1e-12


def test_recognition_validation_loop_matches_oracle(dev, capsys):
    """train_rec.test(): eval-mode forward, CTC mean loss, CER from the device-side greedy decode; previews are printed."""
    import ocrs_models_amd as oa
    from ocrs_models_amd import train_rec as tr
    from oracle import ctc as octc
    from oracle import recognition as orec
    from oracle import text as otext
    from oracle.params import recognition_specs

    m, P, Bf = _load(oa.RecognitionModel(oa.text.DEFAULT_ALPHABET), 33, recognition_specs())
    m = m.to(dev)
    batch = otext.collate(rec_samples(REC_CASE))
    il = (batch["image_width"] // 4).tolist()
    with torch.no_grad():
        lp = orec.forward(P, Bf, batch["image"], False)
        want = float(octc.ctc_loss_torch(lp, batch["text_seq"], il, batch["text_len"].tolist()))
    ostats = otext.AccuracyStats()
    ostats.update(batch["text_seq"], batch["text_len"].tolist(), lp, il)
    loss, stats = tr.test(dev, [batch, batch], m)
    assert not m.training
    assert abs(loss - want) < 1e-4 * abs(want), (loss, want)
    assert stats.total_chars == 2 * ostats.total_chars and stats.char_errors == 2 * ostats.char_errors
    out = capsys.readouterr().out
    assert out.count("Sample test prediction") == batch["image"].shape[0]
