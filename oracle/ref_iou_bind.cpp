// Harness glue (ours, not reference code): exposes the reference's own
// boxes_iou_bev_cpu (det3d/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252, declared in
// iou3d_cpu.h:9) through a plain C entry point so tests can call the compiled
// reference with ctypes.  Built by oracle/build_ref.py into oracle/_ref/.
#include <torch/extension.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

extern "C" int fdref_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out) {
    auto opts = torch::TensorOptions().dtype(torch::kFloat32);
    at::Tensor ta = torch::from_blob(const_cast<float *>(a), {na, 7}, opts);
    at::Tensor tb = torch::from_blob(const_cast<float *>(b), {nb, 7}, opts);
    at::Tensor to = torch::from_blob(out, {na, nb}, opts);
    return boxes_iou_bev_cpu(ta, tb, to);
}
