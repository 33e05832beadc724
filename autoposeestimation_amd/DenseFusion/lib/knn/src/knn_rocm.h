// The ROCm form of the reference's only native interface,
//     int knn(at::Tensor& ref, at::Tensor& query, at::Tensor& idx)          DenseFusion/lib/knn/src/knn.h:12
// as a maintainer of the reference would write it against the C ABI of libape_hip.so (include/ape_hip.h: ape_knn_f32): the
// `#ifdef WITH_CUDA` branch of knn.h:30-50 (workspace allocation + knn_device + error check) becomes ONE call on torch's current
// stream -- no distance matrix, no workspace.  ref [B, D, Nr], query [B, D, Nq], idx [B, k, Nq] int64 (1-based, like knn.cu writes
// them); returns 1 (knn.h:50,63).  This build has no CPU branch: the product path never computes on the host.
#pragma once
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include "ape_hip.h"

inline int knn(at::Tensor& ref, at::Tensor& query, at::Tensor& idx)
{
    TORCH_CHECK(ref.dim() == 3 && query.dim() == 3 && idx.dim() == 3, "knn expects ref[B,D,Nr], query[B,D,Nq], idx[B,k,Nq]");
    TORCH_CHECK(ref.size(0) == query.size(0) && ref.size(1) == query.size(1) && idx.size(0) == ref.size(0) && idx.size(2) == query.size(2),
                "knn: inconsistent shapes");
    TORCH_CHECK(ref.is_cuda() && query.is_cuda() && idx.is_cuda(), "knn (ROCm build): tensors must live on the GPU");
    TORCH_CHECK(ref.scalar_type() == at::kFloat && query.scalar_type() == at::kFloat && idx.scalar_type() == at::kLong,
                "knn: ref / query float32, idx int64");
    TORCH_CHECK(ref.is_contiguous() && query.is_contiguous() && idx.is_contiguous(), "knn: contiguous tensors");
    const int rc = ape_knn_f32(ref.data_ptr<float>(), query.data_ptr<float>(), idx.data_ptr<int64_t>(), (int)ref.size(0), (int)ref.size(1),
                               (int)ref.size(2), (int)query.size(2), (int)idx.size(1), (void*)c10::hip::getCurrentHIPStream().stream());
    TORCH_CHECK(rc == APE_OK, "ape_knn_f32 failed (", rc, "): ", ape_last_error());
    return 1;
}
