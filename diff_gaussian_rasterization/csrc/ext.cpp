// ext.cpp -- the compiled torch extension of the drop-in package `diff_gaussian_rasterization`.
//
// Upstream ships a pybind11 module `_C` with three functions (rasterize_gaussians, rasterize_gaussians_backward,
// mark_visible; SURVEY 8b "C++/HIP extension ABI"; call sites gaussian_renderer/__init__.py:15,89-97).  This file is
// that module for MI355X: torch::Tensor in, torch::Tensor out, the current HIP stream of torch, the three growable
// uint8 scratch tensors handed to the kernels through resize callbacks -- and nothing else: every kernel lives behind
// the plain C ABI of libe3dgs_hip.so (include/e3dgs_hip.h), which this module links against.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>
#include <torch/csrc/autograd/custom_function.h>

#include <functional>
#include <stdexcept>
#include <string>
#include <tuple>

#include "e3dgs_hip.h"

namespace {

using torch::Tensor;

char* resize_cb(void* user, size_t n) {
    Tensor* t = static_cast<Tensor*>(user);
    t->resize_({(long long)n});
    return reinterpret_cast<char*>(t->data_ptr());
}

const float* fptr(const Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

Tensor prep(const Tensor& t, const char* name) {
    if (t.numel() == 0) return t;
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA/HIP tensor (this op has no CPU path)");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
    return t.contiguous();
}

void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + " failed (code " + std::to_string(rc) + "): " + e3dgs_last_error());
}

// `flags`: the C ABI's flags word (0 for upstream's operator; E3DGS_FLAG_PREACT = raw log-scales / quaternions / opacity
// logits with the activations inside the kernels -- what the autograd node below runs for event_3dgs_amd.adopt.render)
std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_forward_impl(
    const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
    const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix,
    const Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height, const int image_width,
    const Tensor& sh, const int degree, const Tensor& campos, const bool prefiltered, const bool debug, const int flags) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    TORCH_CHECK(means3D.is_cuda(), "means3D must be a CUDA/HIP tensor (this op has no CPU path)");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());   // (torch-ROCm tensors say "cuda")
    const int P = (int)means3D.size(0);
    const Tensor m = prep(means3D, "means3D"), col = prep(colors, "colors_precomp"), op = prep(opacity, "opacities"),
                 sc = prep(scales, "scales"), rot = prep(rotations, "rotations"), cov = prep(cov3D_precomp, "cov3D_precomp"),
                 shc = prep(sh, "shs"), bg = prep(background, "bg"), view = prep(viewmatrix, "viewmatrix"),
                 proj = prep(projmatrix, "projmatrix"), cam = prep(campos, "campos");
    const int M = shc.numel() ? (int)shc.size(1) : 0;
    auto fopts = means3D.options().dtype(torch::kFloat32);
    Tensor out_color = torch::empty({3, image_height, image_width}, fopts);      // every pixel is written
    Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
    auto bopts = means3D.options().dtype(torch::kByte);
    Tensor geom = torch::empty({0}, bopts), binning = torch::empty({0}, bopts), img = torch::empty({0}, bopts);
    int rendered = 0;
    check(e3dgs_rasterize_forward(resize_cb, &geom, resize_cb, &binning, resize_cb, &img, P, degree, M, fptr(bg),
                                  image_width, image_height, fptr(m), fptr(shc), fptr(col), fptr(op), fptr(sc),
                                  scale_modifier, fptr(rot), fptr(cov), fptr(view), fptr(proj), fptr(cam), tan_fovx,
                                  tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(), radii.data_ptr<int>(),
                                  debug ? 1 : 0, flags, &rendered, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
          "e3dgs_rasterize_forward");
    return std::make_tuple(rendered, out_color, radii, geom, binning, img);
}

std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians(
    const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
    const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix,
    const Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height, const int image_width,
    const Tensor& sh, const int degree, const Tensor& campos, const bool prefiltered, const bool debug) {
    return rasterize_forward_impl(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                  viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                                  prefiltered, debug, 0);
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_backward_impl(
    const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& colors, const Tensor& scales,
    const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix,
    const Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const Tensor& dL_dout_color, const Tensor& sh,
    const int degree, const Tensor& campos, const Tensor& geomBuffer, const int R, const Tensor& binningBuffer,
    const Tensor& imageBuffer, const bool debug, const int flags, const Tensor& opacity /* PREACT: the logits */) {
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());   // (torch-ROCm tensors say "cuda")
    const int P = (int)means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const Tensor m = prep(means3D, "means3D"), col = prep(colors, "colors_precomp"), sc = prep(scales, "scales"),
                 rot = prep(rotations, "rotations"), cov = prep(cov3D_precomp, "cov3D_precomp"), shc = prep(sh, "shs"),
                 bg = prep(background, "bg"), view = prep(viewmatrix, "viewmatrix"), proj = prep(projmatrix, "projmatrix"),
                 cam = prep(campos, "campos"), g = prep(dL_dout_color, "dL_dout_color"), op = prep(opacity, "opacities");
    const int M = shc.numel() ? (int)shc.size(1) : 0;
    auto o = means3D.options().dtype(torch::kFloat32);
    // groups the call does not differentiate come back as zeros (upstream's convention); the others are fully written
    const bool has_cov = cov.numel() != 0, has_col = col.numel() != 0, has_sh = shc.numel() != 0;
    Tensor dmeans2D = torch::empty({P, 3}, o), dopacity = torch::empty({P, 1}, o), dmeans3D = torch::empty({P, 3}, o);
    Tensor dcolors = has_col ? torch::empty({P, 3}, o) : torch::zeros({P, 3}, o);
    Tensor dcov = has_cov ? torch::empty({P, 6}, o) : torch::zeros({P, 6}, o);
    Tensor dsh = has_sh ? torch::empty({P, M, 3}, o) : torch::zeros({P, M, 3}, o);
    Tensor dscales = has_cov ? torch::zeros({P, 3}, o) : torch::empty({P, 3}, o);
    Tensor drot = has_cov ? torch::zeros({P, 4}, o) : torch::empty({P, 4}, o);
    if (P == 0) return std::make_tuple(dmeans2D, dcolors, dopacity, dmeans3D, dcov, dsh, dscales, drot);
    Tensor acc = torch::empty({(long long)R + P, 12}, o);       // one 48-B record per instance + one sum per Gaussian
    check(e3dgs_rasterize_backward(P, degree, M, R, fptr(bg), W, H, fptr(m), fptr(shc), fptr(col), fptr(op), fptr(sc),
                                   scale_modifier, fptr(rot), fptr(cov), fptr(view), fptr(proj), fptr(cam), tan_fovx,
                                   tan_fovy, radii.data_ptr<int>(), reinterpret_cast<const char*>(geomBuffer.data_ptr()),
                                   reinterpret_cast<const char*>(binningBuffer.data_ptr()),
                                   reinterpret_cast<const char*>(imageBuffer.data_ptr()), fptr(g), acc.data_ptr<float>(),
                                   dmeans2D.data_ptr<float>(), dopacity.data_ptr<float>(),
                                   has_col ? dcolors.data_ptr<float>() : nullptr, dmeans3D.data_ptr<float>(),
                                   has_cov ? dcov.data_ptr<float>() : nullptr, has_sh ? dsh.data_ptr<float>() : nullptr,
                                   has_cov ? nullptr : dscales.data_ptr<float>(), has_cov ? nullptr : drot.data_ptr<float>(),
                                   debug ? 1 : 0, flags, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
          "e3dgs_rasterize_backward");
    return std::make_tuple(dmeans2D, dcolors, dopacity, dmeans3D, dcov, dsh, dscales, drot);
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians_backward(
    const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& colors, const Tensor& scales,
    const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix,
    const Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const Tensor& dL_dout_color, const Tensor& sh,
    const int degree, const Tensor& campos, const Tensor& geomBuffer, const int R, const Tensor& binningBuffer,
    const Tensor& imageBuffer, const bool debug) {
    return rasterize_backward_impl(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                   viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer,
                                   R, binningBuffer, imageBuffer, debug, 0, Tensor());
}

// ---- the autograd node in C++ (upstream keeps it in Python: diff_gaussian_rasterization/__init__.py _RasterizeGaussians).
// Same inputs, same gradient tuple order (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
// cov3D_precomp), same three scratch tensors kept alive between forward and backward -- without a Python frame per
// render() and per backward: the reference's iteration calls the operator three times (train.py:144,159,161).
// "empty tensor = not provided", as in the extension functions above.
struct RasterizeFunction : public torch::autograd::Function<RasterizeFunction> {
    static torch::autograd::variable_list forward(
        torch::autograd::AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D, const Tensor& sh,
        const Tensor& colors, const Tensor& opacities, const Tensor& scales, const Tensor& rotations, const Tensor& cov,
        const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& campos, double scale_modifier,
        double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width, int64_t degree, int64_t flags) {
        (void)means2D;      // only its gradient exists: the NDC-unit screen-space mean gradient (viewspace_points)
        auto r = rasterize_forward_impl(bg, means3D, colors, opacities, scales, rotations, (float)scale_modifier, cov,
                                        viewmatrix, projmatrix, (float)tan_fovx, (float)tan_fovy, (int)image_height,
                                        (int)image_width, sh, (int)degree, campos, false, false, (int)flags);
        const bool preact = (flags & E3DGS_FLAG_PREACT) != 0;
        ctx->save_for_backward({means3D, sh, colors, scales, rotations, cov, bg, viewmatrix, projmatrix, campos,
                                std::get<2>(r), std::get<3>(r), std::get<4>(r), std::get<5>(r),
                                preact ? opacities : Tensor()});
        ctx->saved_data["R"] = (int64_t)std::get<0>(r);
        ctx->saved_data["scale_modifier"] = scale_modifier;
        ctx->saved_data["tan_fovx"] = tan_fovx;
        ctx->saved_data["tan_fovy"] = tan_fovy;
        ctx->saved_data["degree"] = degree;
        ctx->saved_data["flags"] = flags;
        ctx->mark_non_differentiable({std::get<2>(r)});
        return {std::get<1>(r), std::get<2>(r)};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   torch::autograd::variable_list grad_outputs) {
        const auto sv = ctx->get_saved_variables();
        const Tensor &means3D = sv[0], &sh = sv[1], &colors = sv[2], &scales = sv[3], &rotations = sv[4], &cov = sv[5],
                     &bg = sv[6], &view = sv[7], &proj = sv[8], &campos = sv[9], &radii = sv[10], &geom = sv[11],
                     &binning = sv[12], &img = sv[13], &opac = sv[14];
        Tensor g = grad_outputs[0];
        if (!g.defined()) {      // (the image did not reach the loss: a zero pixel gradient of the frame's size)
            const int64_t P = means3D.size(0);
            auto z = [&](std::initializer_list<int64_t> shape) { return torch::zeros(shape, means3D.options().dtype(torch::kFloat32)); };
            const bool hc = cov.defined() && cov.numel() != 0, hcol = colors.defined() && colors.numel() != 0,
                       hsh = sh.defined() && sh.numel() != 0;
            return {z({P, 3}), z({P, 3}), hsh ? torch::zeros_like(sh) : Tensor(), hcol ? z({P, 3}) : Tensor(), z({P, 1}),
                    hc ? Tensor() : z({P, 3}), hc ? Tensor() : z({P, 4}), hc ? z({P, 6}) : Tensor(), Tensor(), Tensor(),
                    Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
        }
        if (g.scalar_type() != torch::kFloat32) g = g.to(torch::kFloat32);
        auto d = rasterize_backward_impl(bg, means3D, radii, colors, scales, rotations,
                                         (float)ctx->saved_data["scale_modifier"].toDouble(), cov, view, proj,
                                         (float)ctx->saved_data["tan_fovx"].toDouble(),
                                         (float)ctx->saved_data["tan_fovy"].toDouble(), g, sh,
                                         (int)ctx->saved_data["degree"].toInt(), campos, geom,
                                         (int)ctx->saved_data["R"].toInt(), binning, img, false,
                                         (int)ctx->saved_data["flags"].toInt(), opac);
        const bool has_cov = cov.defined() && cov.numel() != 0;
        const bool has_col = colors.defined() && colors.numel() != 0, has_sh = sh.defined() && sh.numel() != 0;
        // (dmeans2D, dcolors, dopacity, dmeans3D, dcov, dsh, dscales, drot) -> the operator's input order
        return {std::get<3>(d), std::get<0>(d), has_sh ? std::get<5>(d) : Tensor(), has_col ? std::get<1>(d) : Tensor(),
                std::get<2>(d), has_cov ? Tensor() : std::get<6>(d), has_cov ? Tensor() : std::get<7>(d),
                has_cov ? std::get<4>(d) : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

std::tuple<Tensor, Tensor> rasterize_autograd(const Tensor& means3D, const Tensor& means2D, const Tensor& sh,
                                              const Tensor& colors, const Tensor& opacities, const Tensor& scales,
                                              const Tensor& rotations, const Tensor& cov, const Tensor& bg,
                                              const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& campos,
                                              double scale_modifier, double tan_fovx, double tan_fovy, int64_t image_height,
                                              int64_t image_width, int64_t degree, int64_t flags) {
    auto out = RasterizeFunction::apply(means3D, means2D, sh, colors, opacities, scales, rotations, cov, bg, viewmatrix,
                                        projmatrix, campos, scale_modifier, tan_fovx, tan_fovy, image_height, image_width,
                                        degree, flags);
    return std::make_tuple(out[0], out[1]);
}

Tensor mark_visible(const Tensor& means3D, const Tensor& viewmatrix, const Tensor& projmatrix) {
    const int P = (int)means3D.size(0);
    Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P == 0) return present;
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());   // (torch-ROCm tensors say "cuda")
    const Tensor m = prep(means3D, "means3D"), view = prep(viewmatrix, "viewmatrix"), proj = prep(projmatrix, "projmatrix");
    check(e3dgs_mark_visible(P, fptr(m), fptr(view), fptr(proj), reinterpret_cast<uint8_t*>(present.data_ptr<bool>()),
                             c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
          "e3dgs_mark_visible");
    return present;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
    m.def("mark_visible", &mark_visible);
    m.def("rasterize_autograd", &rasterize_autograd);
    m.def("abi_version", []() { return e3dgs_abi_version(); });
}
