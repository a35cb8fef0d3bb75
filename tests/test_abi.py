"""CPU-only: the C-ABI library builds, loads and exports every symbol include/ocrs_hip.h declares
(no compute calls -- there is no GPU here), and the product path refuses to run without a GPU."""
import ctypes
import os

import pytest
import torch


def test_library_builds_and_exports_every_declared_symbol():
    from ocrs_models_amd import build as b
    from ocrs_models_amd._lib import HEADER_PATH, LIB_PATH, parse_header

    b.build(verbose=False)
    assert os.path.exists(LIB_PATH)
    dll = ctypes.CDLL(LIB_PATH)
    sigs = parse_header(HEADER_PATH)
    assert len(sigs) >= 20
    missing = [n for n in sigs if not hasattr(dll, n)]
    assert not missing, missing


def test_size_queries_work_without_gpu():
    from ocrs_models_amd._lib import lib

    L = lib()
    assert L.opt_chunk() == 2048
    assert L.loss_state_bytes() > 0 and L.loss_hist_bytes() == 2 * 2048 * 4 + 1024 * 2 * 16  # the two histograms + per-block count / sum partials
    assert L.pack_frags_bytes(32, 16, 1) == 64 * 8 * 2
    assert L.pack_frags_bytes(33, 17, 0) == 2 * 2 * 64 * 8 * 4


def test_product_path_has_no_cpu_fallback():
    import ocrs_models_amd as oa

    m = oa.DetectionModel()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 64, 64))
    with pytest.raises(RuntimeError):
        oa.balanced_cross_entropy_loss(torch.rand(1, 1, 4, 4), torch.zeros(1, 1, 4, 4))


def test_state_dict_contract_matches_reference_keys():
    import ocrs_models_amd as oa
    from oracle.params import detection_specs

    sd = oa.DetectionModel().state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(n, s) for n, s, _ in detection_specs()]


def test_bench_byte_model_reads_existing_abi_arguments():
    """bench.py's roofline byte model looks C-ABI call arguments up by name: every name it uses must exist in the header's signature
    of that entry point (an ABI change once shifted positional indices and silently corrupted roofline.achieved)."""
    import bench
    from ocrs_models_amd._lib import ARG_NAMES

    assert set(bench.ALG_BYTES_ARGS) == set(bench.FAMILIES)
    for fam, names in bench.ALG_BYTES_ARGS.items():
        have = ARG_NAMES["ocrs_" + fam]
        for n in names:
            assert n in have, (fam, n, have)
    # and the model returns a positive number for a plausible call of each family
    for fam in bench.FAMILIES:
        args = [8 if n in ("Ca", "Cb", "C", "Cout", "Cup") else (64 if n in ("H", "W", "h", "w") else (2 if n == "N" else (3 if n == "parts" else 0)))
                for n in ARG_NAMES["ocrs_" + fam]]
        # (dw_bwd / bn_bwd_reduce launches add time to their block's backward pass and no bytes: 8(d) books a block backward once)
        assert bench.alg_bytes(fam, args, 2) > 0 or fam in ("dw_bwd", "bn_bwd_reduce"), fam


def test_bench_byte_model_reproduces_survey_totals():
    """sum(in+out) = 277.9 M elements per 1024^2 image (1667 MB bf16 fwd+bwd), 69.5 M at 512^2; 53.35 GB per B=32 step."""
    import bench

    assert abs(bench.det_alg_elems_per_image(1024, 1024) - 277.9e6) < 0.1e6
    assert abs(bench.det_alg_elems_per_image(512, 512) - 69.5e6) < 0.1e6
    assert abs(3 * 2 * bench.det_alg_elems_per_image(1024, 1024) * 32 / 1e9 - 53.35) < 0.01


def test_recognition_state_dict_contract_matches_reference_keys():
    import ocrs_models_amd as oa
    from oracle.params import recognition_specs

    sd = oa.RecognitionModel(oa.text.DEFAULT_ALPHABET).state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(n, s) for n, s, _ in recognition_specs()]


def test_torch_library_ops_are_registered():
    """TORCH_LIBRARY(ocrs, ...) registration over the C ABI builds, loads (no GPU needed) and exposes the schemas."""
    from ocrs_models_amd import build as b
    from ocrs_models_amd import torch_ops

    b.build(verbose=False)
    b.build_torch_ops(verbose=False)
    torch_ops.load()
    assert str(torch.ops.ocrs.head_fwd.default._schema) == "ocrs::head_fwd(Tensor z, Tensor tr, Tensor w, Tensor b) -> Tensor"
    assert "ocrs::ctc_greedy_decode" in str(torch.ops.ocrs.ctc_greedy_decode.default._schema)
    with pytest.raises((RuntimeError, NotImplementedError)):  # registered for the device backend only: no CPU kernel
        torch.ops.ocrs.head_fwd(torch.zeros(1, 2, 2, 8), torch.zeros(3, 8), torch.zeros(8), torch.zeros(1))
