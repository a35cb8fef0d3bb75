// torch.ops.ocrs.* -- PyTorch dispatcher registration (TORCH_LIBRARY) of hot-path entry points over the C ABI of libocrs_hip.so
// ("A PyTorch C++/HIP extension registering ocrs::... via TORCH_LIBRARY(ocrs, m); ... a plain extern "C" launcher layer
// underneath keeps kernels testable without torch").  Each op takes / returns at::Tensors, allocates its outputs with the caching
// allocator, launches on the CURRENT stream of the calling thread (forward: Python main thread, backward: the autograd engine's device
// thread) and turns the C ABI's status codes into TORCH_CHECK failures (= Python RuntimeError, the reference's error convention).
// Host-only translation unit: no device code here, the kernels live in the .hip files behind include/ocrs_hip.h.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include "../../include/ocrs_hip.h"

namespace {

hipStream_t cur_stream() { return at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

int dt_of(const at::Tensor& t) {
    TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16, "ocrs ops take fp32 or bf16 activations");
    return t.scalar_type() == at::kBFloat16 ? 1 : 0;
}
void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed: ", rc == 1 ? "bad argument" : "HIP launch/runtime error"); }

// out_conv: nn.Conv2d(8, 1, 1) + nn.Sigmoid (ocrs_models/models.py:125-129) on an NHWC block output z [N,H,W,8] with its load transform tr [3,8]
at::Tensor head_fwd(const at::Tensor& z, const at::Tensor& tr, const at::Tensor& w, const at::Tensor& b) {
    TORCH_CHECK(z.is_cuda() && z.dim() == 4 && z.size(3) == 8 && z.is_contiguous(), "z must be a contiguous NHWC [N,H,W,8] device tensor");
    TORCH_CHECK(tr.is_cuda() && tr.scalar_type() == at::kFloat && tr.numel() == 24 && w.numel() == 8 && b.numel() == 1, "bad parameter shapes");
    auto pred = at::empty({z.size(0), 1, z.size(1), z.size(2)}, z.options().dtype(at::kFloat));
    check(ocrs_head_fwd(z.data_ptr(), tr.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), pred.data_ptr<float>(),
                        (long)z.size(0) * z.size(1) * z.size(2), dt_of(z), cur_stream()),
          "ocrs::head_fwd");
    return pred;
}

// nn.MaxPool2d(2) over relu(bn(z)) (models.py:54), raw = the selected element's pre-BatchNorm z
at::Tensor maxpool_fwd(const at::Tensor& z, const at::Tensor& tr, bool raw) {
    TORCH_CHECK(z.is_cuda() && z.dim() == 4 && z.is_contiguous() && z.size(3) % 8 == 0, "z must be a contiguous NHWC device tensor, C % 8 == 0");
    auto out = at::empty({z.size(0), z.size(1) / 2, z.size(2) / 2, z.size(3)}, z.options());
    check(ocrs_maxpool_fwd(z.data_ptr(), tr.data_ptr<float>(), out.data_ptr(), (int)z.size(3), (int)z.size(0), (int)z.size(1), (int)z.size(2), raw ? 1 : 0,
                           dt_of(z), cur_stream()),
          "ocrs::maxpool_fwd");
    return out;
}

// preds.argmax(-1) + the greedy CTC collapse (train_rec.py:52; datasets/util.py:147-177): (T,N,C) fp32 log-probs, in_len (N,) int64
std::tuple<at::Tensor, at::Tensor, at::Tensor> ctc_greedy_decode(const at::Tensor& lp, const at::Tensor& in_len) {
    TORCH_CHECK(lp.is_cuda() && lp.dim() == 3 && lp.scalar_type() == at::kFloat && lp.is_contiguous(), "lp must be a contiguous (T,N,C) fp32 device tensor");
    TORCH_CHECK(in_len.is_cuda() && in_len.scalar_type() == at::kLong && in_len.numel() == lp.size(1), "in_len must be (N,) int64 on the device");
    const int T = (int)lp.size(0), N = (int)lp.size(1), C = (int)lp.size(2);
    auto io = lp.options().dtype(at::kInt);
    auto amax = at::empty({N, T}, io), labels = at::zeros({N, T}, io), lens = at::empty({N}, io);
    check(ocrs_ctc_greedy_decode(lp.data_ptr<float>(), (const long long*)in_len.data_ptr<int64_t>(), amax.data_ptr<int>(), labels.data_ptr<int>(),
                                 lens.data_ptr<int>(), T, N, C, cur_stream()),
          "ocrs::ctc_greedy_decode");
    return {amax, labels, lens};
}

}  // namespace

TORCH_LIBRARY(ocrs, m) {
    m.def("head_fwd(Tensor z, Tensor tr, Tensor w, Tensor b) -> Tensor");
    m.def("maxpool_fwd(Tensor z, Tensor tr, bool raw) -> Tensor");
    m.def("ctc_greedy_decode(Tensor lp, Tensor in_len) -> (Tensor, Tensor, Tensor)");
}
TORCH_LIBRARY_IMPL(ocrs, CUDA, m) {
    m.impl("head_fwd", &head_fwd);
    m.impl("maxpool_fwd", &maxpool_fwd);
    m.impl("ctc_greedy_decode", &ctc_greedy_decode);
}
