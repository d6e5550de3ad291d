// INTEGRATION.md section B, compiled: what a maintainer of the reference puts in place of
// submodules/diff-surfel-rasterization/rasterize_points.cu -- a plain C++ translation unit (no CUDA, no GLM, no CUB,
// no hipify) that keeps the pybind surface and forwards to the C ABI of libg4s_hip.so (include/g4s_rasterizer.h).
// torch is plumbing: tensors for memory, the current HIP stream.  Mirrors dsr/rasterize_points.cu:39-254 for shapes,
// checks, zero-fill conventions and tuple orders.
#include "rasterize_points.h"

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <stdexcept>
#include <string>

#include "g4s_rasterizer.h"

namespace {

// resizeFunctional (rasterize_points.cu:31-37) as a function + context pointer
char* resize_cb(void* ctx, size_t n) {
    auto* t = static_cast<torch::Tensor*>(ctx);
    t->resize_({(long long)n});
    return n ? reinterpret_cast<char*>(t->data_ptr()) : reinterpret_cast<char*>(1);  // non-NULL sentinel for empty chunks
}

void check_input(const torch::Tensor& t, const char* name) {  // CHECK_INPUT, rasterize_points.cu:27-28
    if (t.numel() && !t.is_cuda()) throw std::runtime_error(std::string(name) + " must be a CUDA tensor");
}

// contiguous float32 tensor, kept alive by the caller; empty => NULL => "absent" (rasterizer_impl.cu:322-323)
torch::Tensor f32(const torch::Tensor& t) { return t.to(torch::kFloat32).contiguous(); }
const float* fptr(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
void* stream_of(const torch::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
// The tensors' device becomes current for the call: the library launches on the stream it is handed, and the tensors
// allocated here must come from that device's pool (a process with several GPUs may have another one current).
// (the device-generic guard: torch-ROCm registers its HIP guard implementation under the device type "cuda", which is what
// the tensors report; c10::hip::HIPGuard itself refuses that type)
struct DeviceOf {
    c10::OptionalDeviceGuard guard;
    explicit DeviceOf(const torch::Tensor& t) { if (t.is_cuda()) guard.reset_device(t.device()); }
};

}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& transMat_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                       const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                       const bool prefiltered, const bool debug) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    const DeviceOf on_device(means3D);
    check_input(background, "background"); check_input(means3D, "means3D"); check_input(colors, "colors");
    check_input(opacity, "opacity"); check_input(scales, "scales"); check_input(rotations, "rotations");
    check_input(transMat_precomp, "transMat_precomp"); check_input(viewmatrix, "viewmatrix");
    check_input(projmatrix, "projmatrix"); check_input(sh, "sh"); check_input(campos, "campos");
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    const auto fo = means3D.options().dtype(torch::kFloat32);
    const auto bo = torch::TensorOptions(torch::kByte).device(means3D.device());
    torch::Tensor geom = torch::empty({0}, bo), binning = torch::empty({0}, bo), img = torch::empty({0}, bo);
    torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
    if (P == 0)  // rasterize_points.cu:85-99: zero-filled outputs, nothing launched
        return std::make_tuple(0, torch::zeros({3, H, W}, fo), torch::zeros({7, H, W}, fo), radii, geom, binning, img);
    torch::Tensor out_color = torch::empty({3, H, W}, fo), out_others = torch::empty({7, H, W}, fo);  // fully written
    const int M = sh.size(0) != 0 ? (int)sh.size(1) : 0;  // rasterize_points.cu:101-105
    const torch::Tensor bg = f32(background), m3 = f32(means3D), col = f32(colors), opa = f32(opacity), sc = f32(scales),
                        rot = f32(rotations), tm = f32(transMat_precomp), vm = f32(viewmatrix), pm = f32(projmatrix),
                        shc = f32(sh), cp = f32(campos);
    const int rendered = g4s_rasterizer_forward(
        resize_cb, &geom, resize_cb, &binning, resize_cb, &img, P, degree, M, fptr(bg), W, H, fptr(m3), fptr(shc), fptr(col),
        fptr(opa), fptr(sc), scale_modifier, fptr(rot), fptr(tm), fptr(vm), fptr(pm), fptr(cp), tan_fovx, tan_fovy,
        prefiltered ? 1 : 0, out_color.data_ptr<float>(), out_others.data_ptr<float>(), radii.data_ptr<int>(), debug ? 1 : 0,
        stream_of(means3D));
    if (rendered < 0) throw std::runtime_error(g4s_last_error());  // pybind turns it into a Python RuntimeError
    return std::make_tuple(rendered, out_color, out_others, radii, geom, binning, img);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& transMat_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_others,
                               const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                               const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer,
                               const torch::Tensor& imageBuffer, const bool debug) {
    const DeviceOf on_device(means3D);
    check_input(background, "background"); check_input(means3D, "means3D"); check_input(radii, "radii");
    check_input(colors, "colors"); check_input(scales, "scales"); check_input(rotations, "rotations");
    check_input(transMat_precomp, "transMat_precomp"); check_input(viewmatrix, "viewmatrix");
    check_input(projmatrix, "projmatrix"); check_input(sh, "sh"); check_input(campos, "campos");
    check_input(geomBuffer, "geomBuffer"); check_input(binningBuffer, "binningBuffer"); check_input(imageBuffer, "imageBuffer");
    const int P = (int)means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);  // rasterize_points.cu:178-179
    const int M = sh.size(0) != 0 ? (int)sh.size(1) : 0;
    const auto fo = means3D.options().dtype(torch::kFloat32);
    // the library writes every element when P > 0; the reference zero-fills 456 MB here (rasterize_points.cu:187-195)
    auto make = [&](std::initializer_list<int64_t> shape) { return P == 0 ? torch::zeros(shape, fo) : torch::empty(shape, fo); };
    torch::Tensor dL_dmeans3D = make({P, 3}), dL_dmeans2D = make({P, 3}), dL_dcolors = make({P, 3}),
                  dL_dopacity = make({P, 1}), dL_dtransMat = make({P, 9}), dL_dsh = make({P, M, 3}),
                  dL_dscales = make({P, 2}), dL_drotations = make({P, 4});
    if (P != 0) {
        const size_t ws_bytes = g4s_rasterizer_backward_workspace(P, R);
        torch::Tensor workspace = torch::empty({(long long)ws_bytes}, torch::TensorOptions(torch::kByte).device(means3D.device()));
        const torch::Tensor bg = f32(background), m3 = f32(means3D), col = f32(colors), sc = f32(scales), rot = f32(rotations),
                            tm = f32(transMat_precomp), vm = f32(viewmatrix), pm = f32(projmatrix), shc = f32(sh),
                            cp = f32(campos), gc = f32(dL_dout_color), go = f32(dL_dout_others), rad = radii.contiguous();
        const int rc = g4s_rasterizer_backward(
            P, degree, M, R, fptr(bg), W, H, fptr(m3), fptr(shc), fptr(col), fptr(sc), scale_modifier, fptr(rot), fptr(tm),
            fptr(vm), fptr(pm), fptr(cp), tan_fovx, tan_fovy, rad.data_ptr<int>(),
            reinterpret_cast<char*>(geomBuffer.data_ptr()), reinterpret_cast<char*>(binningBuffer.data_ptr()),
            reinterpret_cast<char*>(imageBuffer.data_ptr()), fptr(gc), fptr(go), dL_dmeans2D.data_ptr<float>(),
            /*dL_dnormal (K7 -> K8 scratch, never returned)*/ nullptr, dL_dopacity.data_ptr<float>(),
            dL_dcolors.data_ptr<float>(), dL_dmeans3D.data_ptr<float>(), dL_dtransMat.data_ptr<float>(),
            M ? dL_dsh.data_ptr<float>() : nullptr, dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(),
            reinterpret_cast<char*>(workspace.data_ptr()), ws_bytes, debug ? 1 : 0, stream_of(means3D));
        if (rc != 0) throw std::runtime_error(g4s_last_error());
    }
    // rasterize_points.cu:232
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations);
}

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix) {
    const int P = (int)means3D.size(0);
    const DeviceOf on_device(means3D);
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(at::kBool));  // rasterize_points.cu:240-241
    if (P != 0) {
        check_input(means3D, "means3D");
        const torch::Tensor m3 = f32(means3D), vm = f32(viewmatrix), pm = f32(projmatrix);
        const int rc = g4s_rasterizer_mark_visible(P, fptr(m3), fptr(vm), fptr(pm),
                                                   reinterpret_cast<uint8_t*>(present.data_ptr<bool>()), stream_of(means3D));
        if (rc != 0) throw std::runtime_error(g4s_last_error());
    }
    return present;
}
