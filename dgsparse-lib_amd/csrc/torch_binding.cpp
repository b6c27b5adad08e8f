// torch_binding.cpp -- TORCH_LIBRARY(dgsparse_spmm) over the C ABI (include/dgsparse_hip.h).
//
// The counterpart of the reference's src/spmm.cpp:36-270: the same five ops with the same positional schemas
// (spmm_sum / spmm_max / spmm_min / spmm_mean (rowptr, col, values, colptr, row, csr2csc, dense, has_value,
// algorithm) -> Tensor, csr2csc(rowptr, colind, values) -> Tensor[]) and four torch::autograd::Function classes
// whose backward is one SDDMM (grad of the sparse values) + one SpMM on the CSC arrays (grad of the dense operand).
// Everything below the binding is torch-free: this file only allocates outputs / workspaces with ATen, takes the
// current HIP stream and a device guard (ROCm PyTorch reports its devices as "cuda", hence the *MasqueradingAsCUDA
// flavours of c10::hip's guard/stream; the reference launches on the legacy default stream with no guard,
// src/cuda/spmm_cuda.cu:57), validates arguments with TORCH_CHECK and forwards raw pointers to libdgsparse_hip.so.
//
// Deliberate fixes w.r.t. the reference: mean backward uses 1/deg(source row); the dense gradient is
// built only when needed and always has dense's shape; the values gradient has values' shape.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <mutex>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <vector>

#include "dgsparse_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

void check_rc(int rc, const char *what) { TORCH_CHECK(rc == 0, "dgsparse: ", what, " failed: ", dgs_strerror(rc), " (", rc, ")"); }

Tensor i32vec(const Tensor &t, const char *name) {
  TORCH_CHECK(t.is_cuda(), "dgsparse: ", name, " must live on a GPU (this build has only the HIP back end; no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == at::kInt && t.dim() == 1, "dgsparse: ", name, " must be a 1-D int32 tensor");
  return t.contiguous();
}
Tensor f32mat(const Tensor &t, const char *name) {
  TORCH_CHECK(t.is_cuda(), "dgsparse: ", name, " must live on a GPU (this build has only the HIP back end; no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == at::kFloat && t.dim() == 2, "dgsparse: ", name, " must be a 2-D float32 tensor");
  return t.contiguous();
}
const float *opt_values(const Tensor &values, bool has_value, int64_t nnz, Tensor &keep) {
  if (!has_value) return nullptr;
  TORCH_CHECK(values.is_cuda() && values.scalar_type() == at::kFloat && values.numel() == nnz,
              "dgsparse: values must be a float32 GPU tensor with one entry per stored element");
  keep = values.contiguous().view({-1});
  return keep.data_ptr<float>();
}
dgsStream_t cur_stream() { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }
Tensor workspace(size_t bytes, const Tensor &like) {
  return at::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
}

// C = reduce(A (*) dense); E (arg column ids) is allocated for max/min only.
std::vector<Tensor> spmm_fwd(int op, const Tensor &rowptr_, const Tensor &col_, const Tensor &values, const Tensor &dense_,
                             bool has_value, int64_t algorithm) {
  const Tensor rowptr = i32vec(rowptr_, "rowptr"), col = i32vec(col_, "col"), dense = f32mat(dense_, "dense");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(dense.device());
  const int64_t M = rowptr.numel() - 1, nnz = col.numel(), K = dense.size(0), N = dense.size(1);
  TORCH_CHECK(M >= 0, "dgsparse: rowptr must have at least one element");
  if (N % 4 && N > 4 && dgs_spmm_csr_schedule(op, M, K, (N + 3) & ~int64_t(3), nnz) == DGS_SCHED_PANEL) {
    // dense graph, feature width not a multiple of 4 (e.g. 41 classes): the column-panel schedule needs 16-byte lane
    // vectors, and two small copies buy it.  Feature columns are independent chains: the visible ones are unchanged.
    const Tensor padded = at::constant_pad_nd(dense, {0, ((N + 3) & ~int64_t(3)) - N}, 0);
    auto r = spmm_fwd(op, rowptr, col, values, padded, has_value, algorithm);
    r[0] = r[0].narrow(1, 0, N).contiguous();
    if (r[1].defined()) r[1] = r[1].narrow(1, 0, N).contiguous();
    return r;
  }
  Tensor vkeep;
  const float *vptr = opt_values(values, has_value, nnz, vkeep);
  Tensor out = at::empty({M, N}, dense.options());
  const bool arg = (op == DGS_MAX || op == DGS_MIN);
  Tensor E = arg ? at::empty({M, N}, dense.options().dtype(at::kInt)) : Tensor();
  const size_t wsb = dgs_spmm_csr_workspace_bytes(op, M, N, nnz);
  Tensor ws = wsb ? workspace(wsb, dense) : Tensor();
  check_rc(dgs_spmm_csr_f32(op, M, K, N, nnz, rowptr.data_ptr<int>(), col.data_ptr<int>(), vptr, dense.data_ptr<float>(),
                            out.data_ptr<float>(), arg ? E.data_ptr<int>() : nullptr, (int)algorithm,
                            wsb ? ws.data_ptr() : nullptr, wsb, cur_stream()),
           "spmm");
  return {out, E};
}

Tensor sddmm_impl(const Tensor &rowptr_, const Tensor &col_, const Tensor &D1_, const Tensor &D2_, int op, const Tensor &E) {
  const Tensor rowptr = i32vec(rowptr_, "rowptr"), col = i32vec(col_, "col"), D1 = f32mat(D1_, "D1"), D2 = f32mat(D2_, "D2");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(D1.device());
  const int64_t M = rowptr.numel() - 1, nnz = col.numel(), F = D1.size(1);
  TORCH_CHECK(D2.size(1) == F && D1.size(0) >= M, "dgsparse: sddmm shape mismatch");
  if (F % 4 && F > 4 &&
      dgs_sddmm_csr_schedule(M, D2.size(0), (F + 3) & ~int64_t(3), nnz, E.defined() ? 1 : 0) == DGS_SCHED_PANEL) {
    // same trick as in spmm_fwd: zero feature columns add exact zeros to every dot product
    const int64_t pad = ((F + 3) & ~int64_t(3)) - F;
    Tensor Ep = E.defined() ? at::constant_pad_nd(E, {0, pad}, -1) : Tensor();
    return sddmm_impl(rowptr, col, at::constant_pad_nd(D1, {0, pad}, 0), at::constant_pad_nd(D2, {0, pad}, 0), op, Ep);
  }
  Tensor out = at::empty({nnz}, D1.options());
  if (E.defined()) {
    TORCH_CHECK(E.scalar_type() == at::kInt && E.sizes() == D1.sizes(), "dgsparse: E must be int32 with the shape of D1");
    const Tensor Ec = E.contiguous();
    check_rc(dgs_sddmm_csr_mask_f32(M, D2.size(0), F, nnz, rowptr.data_ptr<int>(), col.data_ptr<int>(), D1.data_ptr<float>(),
                                    D2.data_ptr<float>(), Ec.data_ptr<int>(), out.data_ptr<float>(), cur_stream()),
             "sddmm_mask");
  } else {
    check_rc(dgs_sddmm_csr_f32(op, M, D2.size(0), F, nnz, rowptr.data_ptr<int>(), col.data_ptr<int>(), D1.data_ptr<float>(),
                               D2.data_ptr<float>(), out.data_ptr<float>(), cur_stream()),
             "sddmm");
  }
  return out;
}

Tensor spmm_mask_impl(const Tensor &ptr_, const Tensor &idx_, const Tensor &tvalues, bool has_value, const Tensor &grad_,
                      const Tensor &E, int64_t n_out) {
  const Tensor ptr = i32vec(ptr_, "colptr"), idx = i32vec(idx_, "row"), grad = f32mat(grad_, "grad");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(grad.device());
  const int64_t Mo = ptr.numel() - 1, nnz = idx.numel(), Mi = grad.size(0), N = grad.size(1);
  Tensor vkeep;
  const float *vptr = opt_values(tvalues, has_value, nnz, vkeep);
  const Tensor Ec = E.contiguous();
  const int64_t rows = std::max(n_out, Mo);
  Tensor out = at::empty({rows, N}, grad.options());
  if (rows > Mo) out.narrow(0, Mo, rows - Mo).zero_();
  const size_t wsb = dgs_spmm_csr_mask_workspace_bytes(Mo, N, nnz);
  Tensor ws = wsb ? workspace(wsb, grad) : Tensor();
  check_rc(dgs_spmm_csr_mask_f32(Mo, Mi, N, nnz, ptr.data_ptr<int>(), idx.data_ptr<int>(), vptr, grad.data_ptr<float>(),
                                 Ec.data_ptr<int>(), out.data_ptr<float>(), wsb ? ws.data_ptr() : nullptr, wsb, cur_stream()),
           "spmm_mask");
  return out;
}

// values in CSC order = values[csr2csc]: one pass of the HIP gather over the int32 permutation (index_select wants an
// int64 copy of the permutation first and takes 2.4 ms for the 114.6 M entries of a Reddit-sized graph; this, 1.9 ms -
// a random 4-byte gather).  The last result is kept: every layer of a model that shares one adjacency asks for the same
// permuted values in its backward, and with fixed edge weights every iteration does.  A hit needs the SAME tensor
// object (weak reference to its TensorImpl, so a recycled address cannot alias), the same version counter (any
// in-place update since then misses - the trust model of autograd's own saved-tensor check) and the same permutation.
struct TValuesCache {
  c10::weak_intrusive_ptr<c10::TensorImpl> impl{c10::intrusive_ptr<c10::TensorImpl>()};
  uint32_t version = 0;
  const void *perm = nullptr;
  Tensor out;
};
// heap-allocated and never destroyed: a static Tensor would be freed by a static destructor at process exit, after the
// HIP runtime and the caching allocator may already be gone
static TValuesCache &g_tv = *new TValuesCache();
static std::mutex g_tv_mu;

Tensor t_values(const Tensor &values, const Tensor &csr2csc, bool has_value) {
  if (!has_value) return Tensor();
  if (csr2csc.scalar_type() != at::kInt || !csr2csc.is_cuda())
    return values.view({-1}).index_select(0, csr2csc.to(at::kLong));
  const Tensor perm = csr2csc.contiguous();
  // never inside a stream capture: a hit would leave the gather out of the graph, and replays would read stale values
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(static_cast<hipStream_t>(cur_stream()), &cap) != hipSuccess ||
                         cap != hipStreamCaptureStatusNone;
  if (!capturing) {
    std::lock_guard<std::mutex> lk(g_tv_mu);
    if (g_tv.out.defined() && g_tv.perm == perm.data_ptr() && g_tv.version == values._version() &&
        g_tv.out.numel() == perm.numel()) {
      const auto alive = g_tv.impl.lock();
      if (alive && alive.get() == values.unsafeGetTensorImpl()) return g_tv.out;
    }
  }
  Tensor vkeep;
  const float *vptr = opt_values(values, true, perm.numel(), vkeep);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(vkeep.device());
  Tensor out = at::empty({perm.numel()}, vkeep.options());
  check_rc(dgs_gather_rows_f32(perm.numel(), 1, perm.data_ptr<int>(), vptr, out.data_ptr<float>(), cur_stream()),
           "gather");
  if (!capturing) {
    std::lock_guard<std::mutex> lk(g_tv_mu);
    g_tv.impl = c10::weak_intrusive_ptr<c10::TensorImpl>(values.getIntrusivePtr());
    g_tv.version = values._version();
    g_tv.perm = perm.data_ptr();
    g_tv.out = out;
  }
  return out;
}
Tensor pad_rows(const Tensor &g, int64_t n) {
  if (g.size(0) == n) return g;
  Tensor out = at::zeros({n, g.size(1)}, g.options());
  out.narrow(0, 0, g.size(0)).copy_(g);
  return out;
}

// ---- autograd ------------------------------------------------------------------------------------------
template <int OP>
struct SpMM : public torch::autograd::Function<SpMM<OP>> {
  static Tensor forward(AutogradContext *ctx, Tensor rowptr, Tensor col, Tensor values, Tensor colptr, Tensor row,
                        Tensor csr2csc, Tensor dense, bool has_value, int64_t algorithm) {
    auto out = spmm_fwd(OP, rowptr, col, values, dense, has_value, algorithm);
    ctx->saved_data["has_value"] = has_value;
    ctx->saved_data["algorithm"] = algorithm;
    if (OP == DGS_MAX || OP == DGS_MIN)
      ctx->save_for_backward({rowptr, col, values, colptr, row, csr2csc, dense, out[1]});
    else
      ctx->save_for_backward({rowptr, col, values, colptr, row, csr2csc, dense});
    return out[0];
  }

  static tensor_list backward(AutogradContext *ctx, tensor_list grad_outs) {
    const Tensor grad_out = grad_outs[0].contiguous();
    const bool has_value = ctx->saved_data["has_value"].toBool();
    const int64_t algorithm = ctx->saved_data["algorithm"].toInt();
    const auto saved = ctx->get_saved_variables();
    const Tensor &rowptr = saved[0], &col = saved[1], &values = saved[2], &colptr = saved[3], &row = saved[4],
                 &csr2csc = saved[5], &dense = saved[6];
    Tensor grad_value, grad_dense;
    const bool need_v = has_value && ctx->needs_input_grad(2), need_d = ctx->needs_input_grad(6);
    if (OP == DGS_MAX || OP == DGS_MIN) {
      const Tensor &E = saved[7];
      if (!at::globalContext().deterministicAlgorithms() && (need_v || need_d)) {
        // one pass over the arg ids, fp32 atomics (not bit-reproducible run to run; the masked kernels below are)
        const Tensor rp = i32vec(rowptr, "rowptr"), cl = i32vec(col, "col"), Xc = f32mat(dense, "dense");
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(Xc.device());
        const int64_t M = rp.numel() - 1, nnz = cl.numel(), K = Xc.size(0), N = Xc.size(1);
        Tensor vkeep;
        const float *vptr = opt_values(values, has_value, nnz, vkeep);
        const Tensor Ec = E.contiguous();
        if (need_d) grad_dense = at::empty({K, N}, Xc.options());
        Tensor gw = need_v ? at::empty({nnz}, Xc.options()) : Tensor();
        check_rc(dgs_spmm_arg_backward_f32(M, K, N, nnz, rp.data_ptr<int>(), cl.data_ptr<int>(), vptr, Ec.data_ptr<int>(),
                                           grad_out.data_ptr<float>(), Xc.data_ptr<float>(),
                                           need_d ? grad_dense.data_ptr<float>() : nullptr,
                                           need_v ? gw.data_ptr<float>() : nullptr, cur_stream()),
                 "spmm_arg_backward");
        if (need_v) grad_value = gw.view_as(values);
        return {Tensor(), Tensor(), grad_value, Tensor(), Tensor(), Tensor(), grad_dense, Tensor(), Tensor()};
      }
      if (need_v) grad_value = sddmm_impl(rowptr, col, grad_out, dense, DGS_SUM, E).view_as(values);
      if (need_d) grad_dense = spmm_mask_impl(colptr, row, t_values(values, csr2csc, has_value), has_value, grad_out, E, dense.size(0));
    } else if (OP == DGS_MEAN) {
      if (need_v) grad_value = sddmm_impl(rowptr, col, grad_out, dense, DGS_MEAN, Tensor()).view_as(values);
      if (need_d) {  // A^T diag(1/deg) dC: scale grad rows by 1/deg(source row), then a plain transposed SpMM
        const Tensor deg = (rowptr.slice(0, 1) - rowptr.slice(0, 0, -1)).clamp_min(1).to(at::kFloat);
        grad_dense = pad_rows(spmm_fwd(DGS_SUM, colptr, row, t_values(values, csr2csc, has_value), grad_out / deg.unsqueeze(1), has_value, algorithm)[0], dense.size(0));
      }
    } else {
      if (need_v) grad_value = sddmm_impl(rowptr, col, grad_out, dense, DGS_SUM, Tensor()).view_as(values);
      if (need_d) grad_dense = pad_rows(spmm_fwd(DGS_SUM, colptr, row, t_values(values, csr2csc, has_value), grad_out, has_value, algorithm)[0], dense.size(0));
    }
    return {Tensor(), Tensor(), grad_value, Tensor(), Tensor(), Tensor(), grad_dense, Tensor(), Tensor()};
  }
};

template <int OP>
Tensor spmm_op(Tensor rowptr, Tensor col, Tensor values, Tensor colptr, Tensor row, Tensor csr2csc, Tensor dense,
               bool has_value, int64_t algorithm) {
  return SpMM<OP>::apply(rowptr, col, values, colptr, row, csr2csc, dense, has_value, algorithm);
}

// csr2csc(rowptr, colind, values) -> [colptr, row, values in CSC order]; square like the reference (src/spmm.cpp:91-94)
std::vector<Tensor> csr2csc_op(Tensor rowptr_, Tensor colind_, Tensor values) {
  const Tensor rowptr = i32vec(rowptr_, "rowptr"), col = i32vec(colind_, "colind");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(rowptr.device());
  const int64_t n = rowptr.numel() - 1, nnz = col.numel();
  Tensor vkeep;
  const float *vptr = opt_values(values, true, nnz, vkeep);
  Tensor colptr = at::empty({n + 1}, rowptr.options()), row = at::empty({nnz}, rowptr.options());
  Tensor cscval = at::empty({nnz}, vkeep.options());
  const size_t wsb = dgs_csr2csc_workspace_bytes(n, n, nnz);
  Tensor ws = workspace(wsb, vkeep);
  check_rc(dgs_csr2csc_i32(n, n, nnz, rowptr.data_ptr<int>(), col.data_ptr<int>(), vptr, colptr.data_ptr<int>(),
                           row.data_ptr<int>(), cscval.data_ptr<float>(), nullptr, ws.data_ptr(), wsb, cur_stream()),
           "csr2csc");
  return {colptr, row, cscval};
}

// transpose with an explicit column count + the integer permutation (what Storage needs)
std::vector<Tensor> csr2csc_perm_op(Tensor rowptr_, Tensor colind_, int64_t n_cols) {
  const Tensor rowptr = i32vec(rowptr_, "rowptr"), col = i32vec(colind_, "colind");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(rowptr.device());
  const int64_t M = rowptr.numel() - 1, nnz = col.numel();
  Tensor colptr = at::empty({n_cols + 1}, rowptr.options()), row = at::empty({nnz}, rowptr.options()),
         perm = at::empty({nnz}, rowptr.options());
  const size_t wsb = dgs_csr2csc_workspace_bytes(M, n_cols, nnz);
  Tensor ws = workspace(wsb, rowptr);
  check_rc(dgs_csr2csc_i32(M, n_cols, nnz, rowptr.data_ptr<int>(), col.data_ptr<int>(), nullptr, colptr.data_ptr<int>(),
                           row.data_ptr<int>(), nullptr, perm.data_ptr<int>(), ws.data_ptr(), wsb, cur_stream()),
           "csr2csc");
  return {colptr, row, perm};
}

Tensor sddmm_op(Tensor rowptr, Tensor col, Tensor D1, Tensor D2, int64_t reduce_op) {
  return sddmm_impl(rowptr, col, D1, D2, (int)reduce_op, Tensor());
}

// inference entry: no autograd node, returns [out, E?]
std::vector<Tensor> spmm_raw_op(int64_t op, Tensor rowptr, Tensor col, Tensor values, Tensor dense, bool has_value,
                                int64_t algorithm) {
  TORCH_CHECK(op >= 0 && op <= 3, "dgsparse: bad reduce op");
  return spmm_fwd((int)op, rowptr, col, values, dense, has_value, algorithm);
}

}  // namespace

TORCH_LIBRARY(dgsparse_spmm, m) {
  m.def("spmm_sum", &spmm_op<DGS_SUM>);
  m.def("spmm_max", &spmm_op<DGS_MAX>);
  m.def("spmm_min", &spmm_op<DGS_MIN>);
  m.def("spmm_mean", &spmm_op<DGS_MEAN>);
  m.def("csr2csc", &csr2csc_op);
  // additions
  m.def("sddmm", &sddmm_op);
  m.def("csr2csc_perm", &csr2csc_perm_op);
  m.def("spmm_raw", &spmm_raw_op);
}
