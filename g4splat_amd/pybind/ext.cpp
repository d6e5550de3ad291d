// pybind module of the INTEGRATION.md section B route: the reference's module surface (dsr/ext.cpp:15-19 -- three
// functions under these names) over the C-ABI stub in rasterize_points.cpp.
#include "rasterize_points.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("rasterize_gaussians", &RasterizeGaussiansCUDA, "forward through g4s_rasterizer_forward");
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackwardCUDA, "backward through g4s_rasterizer_backward");
    m.def("mark_visible", &markVisible, "near-plane visibility through g4s_rasterizer_mark_visible");
}
