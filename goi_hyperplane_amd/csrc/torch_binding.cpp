// Compiled torch binding of libgoi_raster.so: the C++ file a maintainer of the reference would put in place of
// submodules/diff-gaussian-rasterization/rasterize_points.cu (+ ext.cpp:15-20).  HOST C++ ONLY -- no kernels, no
// hipify: it allocates tensors, fills a GoiRasterScene and calls the C ABI of include/goi_raster.h on torch's current
// HIP stream.  The four functions of the reference's pybind module keep their names, argument order, "empty tensor =
// absent" convention and return tuples:
//
//   rasterize_gaussians            <- RasterizeGaussiansCUDA          rasterize_points.cu:35-123
//   rasterize_gaussians_backward   <- RasterizeGaussiansBackwardCUDA  rasterize_points.cu:213-306
//   rasterize_gaussians_trace      <- RasterizeGaussiansTraceCUDA     rasterize_points.cu:125-211
//   mark_visible                   <- markVisible                     rasterize_points.cu:308-327
//
// plus what this build adds behind the same boundary (goi_hyperplane_amd/_C.py uses them when the module is built):
// the speculative forward (goi_raster_forward_async), the backward with optional upstream gradients / factored dL/dSH
// / gradients as views of one flat buffer, and the feature-gradient-only backward.
//
// Built by goi_hyperplane_amd/build.py (g++ against the torch headers, linked with libgoi_raster.so, rpath $ORIGIN).
#include <torch/extension.h>

#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "goi_raster.h"

namespace {

using torch::Tensor;

void* grow(void* user, size_t bytes) {  // reference: resizeFunctional, rasterize_points.cu:27-33
    auto* t = static_cast<Tensor*>(user);
    // num_rendered changes with every view: 16 MiB steps let the caching allocator hand back the same block
    const long long step = 16ll << 20;
    t->resize_({((long long)bytes + step - 1) / step * step});
    return t->data_ptr();
}

// empty tensor => NULL (rasterize_points.cu:98-111); everything else must be a contiguous fp32 tensor on `dev`
struct Arg {
    Tensor keep;
    const float* p = nullptr;
};
Arg arg(const Tensor& t, const char* name, const c10::Device& dev) {
    Arg a;
    if (!t.defined() || t.numel() == 0) return a;
    TORCH_CHECK_TYPE(t.scalar_type() == torch::kFloat32, name, " must be torch.float32, got ", t.scalar_type());
    TORCH_CHECK_VALUE(t.device() == dev, name, " is on ", t.device(), ", expected ", dev);
    a.keep = t.contiguous();
    a.p = a.keep.data_ptr<float>();
    return a;
}
Arg arg(const c10::optional<Tensor>& t, const char* name, const c10::Device& dev) {
    return t.has_value() ? arg(*t, name, dev) : Arg();
}

c10::Device check_device(const Tensor& means3D) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");  // :58-60
    TORCH_CHECK(means3D.is_cuda(), "goi_hyperplane_amd: tensors must live on a ROCm GPU (cuda device); there is no CPU "
                                   "fallback in this package");
    return means3D.device();
}

void* stream_of(const c10::Device& dev) { return c10::hip::getCurrentHIPStream(dev.index()).stream(); }

[[noreturn]] void raise_last() { throw std::runtime_error(goi_raster_last_error()); }

int semantic_channels(const Tensor& semantics, int P) {
    if (!semantics.defined() || semantics.numel() == 0) {
        TORCH_CHECK(P == 0, "semantics [P,S] is required (the reference dereferences it unconditionally, "
                            "cuda_rasterizer/forward.cu:363)");
        return 10;
    }
    TORCH_CHECK(semantics.ndimension() == 2 && semantics.size(0) == P, "semantics must have dimensions (num_points, S)");
    const int S = (int)semantics.size(1);
    TORCH_CHECK(S >= 1 && S <= 32, "unsupported number of semantic channels S=", S, " (1..32)");
    return S;
}

GoiRasterScene scene_of(int P, int degree, const Tensor& sh, int S, int W, int H, const Arg& bg, const Arg& means3D,
                        const Arg& shs, const Arg& colors, const Arg& semantics, const Arg& opacity, const Arg& scales,
                        float scale_modifier, const Arg& rotations, const Arg& cov3D, const Arg& view, const Arg& proj,
                        const Arg& campos, float tan_fovx, float tan_fovy, bool prefiltered, bool debug) {
    GoiRasterScene sc;
    sc.P = P;
    sc.D = degree;
    sc.M = (sh.defined() && sh.numel()) ? (int)sh.size(1) : 0;
    sc.S = S;
    sc.W = W;
    sc.H = H;
    sc.bg = bg.p;
    sc.means3D = means3D.p;
    sc.shs = shs.p;
    sc.colors_precomp = colors.p;
    sc.semantics = semantics.p;
    sc.opacities = opacity.p;
    sc.scales = scales.p;
    sc.scale_modifier = scale_modifier;
    sc.rotations = rotations.p;
    sc.cov3D_precomp = cov3D.p;
    sc.viewmatrix = view.p;
    sc.projmatrix = proj.p;
    sc.campos = campos.p;
    sc.tan_fovx = tan_fovx;
    sc.tan_fovy = tan_fovy;
    sc.prefiltered = prefiltered ? 1 : 0;
    sc.debug = debug ? 1 : 0;
    return sc;
}

// ---- forward --------------------------------------------------------------------------------------------------------
#define GOI_FORWARD_ARGS                                                                                               \
    const Tensor &background, const Tensor &means3D, const Tensor &colors, const Tensor &semantics,                    \
        const Tensor &opacity, const Tensor &scales, const Tensor &rotations, const float scale_modifier,              \
        const Tensor &cov3D_precomp, const Tensor &viewmatrix, const Tensor &projmatrix, const float tan_fovx,         \
        const float tan_fovy, const int image_height, const int image_width, const Tensor &sh, const int degree,       \
        const Tensor &campos, const bool prefiltered, const bool debug

struct Prepared {
    c10::Device dev;
    int P, H, W, S;
    Arg a[12];
    GoiRasterScene sc;
};

Prepared prepare(GOI_FORWARD_ARGS, bool need_semantics) {
    Prepared f{check_device(means3D)};
    f.P = (int)means3D.size(0);
    f.H = image_height;
    f.W = image_width;
    f.S = need_semantics ? semantic_channels(semantics, f.P) : (int)semantics.size(0);
    const auto& dev = f.dev;
    f.a[0] = arg(background, "background", dev);
    f.a[1] = arg(means3D, "means3D", dev);
    f.a[2] = arg(sh, "sh", dev);
    f.a[3] = arg(colors, "colors_precomp", dev);
    f.a[4] = need_semantics ? arg(semantics, "semantics", dev) : Arg();
    f.a[5] = arg(opacity, "opacities", dev);
    f.a[6] = arg(scales, "scales", dev);
    f.a[7] = arg(rotations, "rotations", dev);
    f.a[8] = arg(cov3D_precomp, "cov3D_precomp", dev);
    f.a[9] = arg(viewmatrix, "viewmatrix", dev);
    f.a[10] = arg(projmatrix, "projmatrix", dev);
    f.a[11] = arg(campos, "campos", dev);
    f.sc = scene_of(f.P, degree, sh, f.S, f.W, f.H, f.a[0], f.a[1], f.a[2], f.a[3], f.a[4], f.a[5], f.a[6], scale_modifier,
                    f.a[7], f.a[8], f.a[9], f.a[10], f.a[11], tan_fovx, tan_fovy, prefiltered, debug);
    return f;
}

std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians(GOI_FORWARD_ARGS) {
    Prepared f = prepare(background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                         viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                         prefiltered, debug, true);
    c10::hip::HIPGuard guard(f.dev.index());
    auto f32 = means3D.options().dtype(torch::kFloat32);
    auto bytes = means3D.options().dtype(torch::kByte);
    Tensor out_color = torch::empty({3, f.H, f.W}, f32), out_sem = torch::empty({f.S, f.H, f.W}, f32);
    Tensor out_depth = torch::empty({1, f.H, f.W}, f32), out_alpha = torch::empty({1, f.H, f.W}, f32);
    Tensor radii = torch::empty({f.P}, means3D.options().dtype(torch::kInt32));
    Tensor geom = torch::empty({f.P > 0 ? (long long)goi_raster_geom_bytes(f.P) : 0}, bytes);
    Tensor img = torch::empty({f.P > 0 ? (long long)goi_raster_image_bytes(f.W, f.H) : 0}, bytes);
    Tensor binning = torch::empty({0}, bytes);
    const int n = goi_raster_forward(&f.sc, f.P ? geom.data_ptr() : nullptr, f.P ? img.data_ptr() : nullptr, grow, &binning,
                                     out_color.data_ptr<float>(), out_sem.data_ptr<float>(), out_depth.data_ptr<float>(),
                                     out_alpha.data_ptr<float>(), f.P ? radii.data_ptr<int>() : nullptr, stream_of(f.dev));
    if (n < 0) raise_last();
    return std::make_tuple(n, out_color, out_sem, out_depth, out_alpha, radii, geom, binning, img);
}

// speculative forward: nothing waits; returns the read-back ticket instead of num_rendered (include/goi_raster.h)
// zcut_in / zcut_out: the speculative depth cut-off of the tile lists (goi_raster_forward_async_cut): per-tile float32 arrays on
// the device, either may be absent (None)
std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians_async(
    GOI_FORWARD_ARGS, const int capacity, const c10::optional<Tensor>& zcut_in, const c10::optional<Tensor>& zcut_out) {
    Prepared f = prepare(background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                         viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                         prefiltered, debug, true);
    c10::hip::HIPGuard guard(f.dev.index());
    auto f32 = means3D.options().dtype(torch::kFloat32);
    auto bytes = means3D.options().dtype(torch::kByte);
    Tensor out_color = torch::empty({3, f.H, f.W}, f32), out_sem = torch::empty({f.S, f.H, f.W}, f32);
    Tensor out_depth = torch::empty({1, f.H, f.W}, f32), out_alpha = torch::empty({1, f.H, f.W}, f32);
    Tensor radii = torch::empty({f.P}, means3D.options().dtype(torch::kInt32));
    Tensor geom = torch::empty({(long long)goi_raster_geom_bytes(f.P)}, bytes);
    Tensor img = torch::empty({(long long)goi_raster_image_bytes(f.W, f.H)}, bytes);
    const long long step = 16ll << 20;
    Tensor binning = torch::empty({((long long)goi_raster_binning_bytes(capacity) + step - 1) / step * step}, bytes);
    const long long tiles = ((long long)(f.W + 15) / 16) * ((f.H + 15) / 16);
    const float* zin = nullptr;
    float* zout = nullptr;
    for (int k = 0; k < 2; k++) {
        const auto& z = k == 0 ? zcut_in : zcut_out;
        if (!z.has_value() || !z->defined()) continue;
        TORCH_CHECK(z->scalar_type() == torch::kFloat32 && z->is_contiguous() && z->device() == f.dev && z->numel() == tiles,
                    "zcut arrays must be contiguous float32 [tiles] tensors on the frame's device");
        if (k == 0) zin = z->data_ptr<float>();
        else zout = z->data_ptr<float>();
    }
    const int ticket = goi_raster_forward_async_cut(&f.sc, geom.data_ptr(), img.data_ptr(), binning.data_ptr(), capacity,
                                                    out_color.data_ptr<float>(), out_sem.data_ptr<float>(),
                                                    out_depth.data_ptr<float>(), out_alpha.data_ptr<float>(),
                                                    radii.data_ptr<int>(), zin, zout, stream_of(f.dev));
    if (ticket < 0) raise_last();
    return std::make_tuple(ticket, out_color, out_sem, out_depth, out_alpha, radii, geom, binning, img);
}

// ---- backward -------------------------------------------------------------------------------------------------------
// grow-only scratch per (device, stream): 4*R*129 bytes, dead when the call returns (same stream)
std::mutex g_scratch_mu;
std::map<std::pair<int, void*>, Tensor> g_scratch;

void* backward_scratch(size_t bytes, const c10::Device& dev, void* stream, const torch::TensorOptions& byte_opts) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    auto key = std::make_pair((int)dev.index(), stream);
    auto it = g_scratch.find(key);
    if (it == g_scratch.end() || (size_t)it->second.numel() < bytes) {
        if (it != g_scratch.end()) g_scratch.erase(it);  // free before growing
        g_scratch[key] = torch::empty({(long long)(bytes + bytes / 4) + 256}, byte_opts);
        it = g_scratch.find(key);
    }
    return it->second.data_ptr();
}

// ---- gradient-buffer pool (goi_raster_backward2: rows that already hold zeros are not written again) -------------------------
// The reference hands autograd eleven freshly zero-filled [P, ..] tensors per backward (rasterize_points.cu:252-262); on the
// headline scene half of the Gaussians are invisible in any one view: 170 MB of zeros per step.  The binding keeps the ONE
// allocation all outputs of a backward are views of, together with that frame's radii, and hands it out again once (a) nobody
// else holds it any more (the storage's reference count is back to the pool's own) and (b) nothing has written to it in place
// (the version counter its views share is where the backward left it: an in-place collective, a gradient clip, zero_() all
// bump it -- such a buffer is reused as if it were fresh).  The kernel then skips the rows of Gaussians that were invisible
// then and are invisible now.  Keyed by device, stream and layout; a few buffers per key (a loop that keeps the gradients of
// step k alive while step k + 1 runs alternates between two).  GOI_GRAD_POOL=0 / set_grad_pool(false) turns it off.
struct PoolEntry {
    Tensor all;         // every output of one backward is a view of this
    Tensor prev_radii;  // radii of the backward that last wrote it
    int64_t version;    // version counter of `all` when that backward returned
};
struct PoolKey {
    int dev;
    void* stream;
    long long total;
    int P, M, S;
    bool operator<(const PoolKey& o) const {
        return std::tie(dev, stream, total, P, M, S) < std::tie(o.dev, o.stream, o.total, o.P, o.M, o.S);
    }
};
std::mutex g_pool_mu;
std::map<PoolKey, std::vector<PoolEntry>> g_pool;
bool g_pool_on = []() {
    const char* e = std::getenv("GOI_GRAD_POOL");
    return !(e && (e[0] == '0' || e[0] == 'n' || e[0] == 'f'));
}();
long long g_pool_hits = 0, g_pool_dirty = 0, g_pool_fresh = 0;
constexpr size_t POOL_BUFFERS_PER_KEY = 3;

void set_grad_pool(bool on) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_on = on;
    if (!on) g_pool.clear();
}
std::tuple<long long, long long, long long> grad_pool_stats() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    return std::make_tuple(g_pool_hits, g_pool_dirty, g_pool_fresh);
}

long long release_scratch() {
    long long freed = 0;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        for (auto& kv : g_scratch) freed += kv.second.numel();
        g_scratch.clear();
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& kv : g_pool)
        for (auto& e : kv.second) freed += e.all.numel() * 4;
    g_pool.clear();
    return freed;
}

// Full backward.  Upstream gradients may be absent (None: the loss does not use that output).  sh_factored: dL/dSH is
// not formed (returned undefined) and dL_dcolors is the clamp-masked colour gradient.  The six parameter gradients are
// views of ONE flat buffer, 256-byte aligned sections, so a data-parallel exchange is a single all-reduce (dist.py).
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> backward_ex(
    const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& colors, const Tensor& semantics,
    const Tensor& scales, const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp,
    const Tensor& viewmatrix, const Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
    const c10::optional<Tensor>& dL_dout_color, const c10::optional<Tensor>& dL_dout_semantic,
    const c10::optional<Tensor>& dL_dout_depth, const c10::optional<Tensor>& dL_dout_alpha, const Tensor& sh,
    const int degree, const Tensor& campos, const Tensor& geomBuffer, const int R, const Tensor& binningBuffer,
    const Tensor& imageBuffer, const Tensor& alphas, const bool debug, const bool sh_factored_in,
    const int scratch_instances /* 0: the row scratch is laid out for R (goi_raster_backward3) */,
    const c10::optional<Tensor>& accumulate_into /* the dL_dmeans3D an earlier call of the same batch returned: this view's
                                                    gradients are ADDED to that call's buffer (GOI_BACKWARD_ACCUMULATE) */) {
    const c10::Device dev = check_device(means3D);
    c10::hip::HIPGuard guard(dev.index());
    const int P = (int)means3D.size(0);
    const int H = (int)alphas.size(-2), W = (int)alphas.size(-1);
    const int S = (int)semantics.size(1);
    const int M = (sh.defined() && sh.numel()) ? (int)sh.size(1) : 0;
    const bool sh_factored = sh_factored_in && M > 0;
    auto f32 = means3D.options().dtype(torch::kFloat32);
    // sections of ONE allocation: the six parameter gradients first (the span a data-parallel exchange reduces), then the other
    // per-Gaussian outputs of the call
    const long long sizes[11] = {3ll * P, sh_factored ? 0ll : 3ll * M * P, (long long)S * P, (long long)P, 3ll * P, 4ll * P,
                                 3ll * P, 3ll * P, (long long)P, 4ll * P, 6ll * P};
    long long offs[11], total = 0;
    for (int i = 0; i < 11; i++) {
        offs[i] = total;
        total += (sizes[i] + 63) / 64 * 64;
    }
    void* stream = stream_of(dev);
    // a pooled buffer nobody else holds any more (see the pool's comment), or a fresh one
    Tensor flat, prev_radii;
    const PoolKey key{(int)dev.index(), stream, total, P, sh_factored ? -M : M, S};
    const bool accumulate = accumulate_into.has_value() && accumulate_into->defined() && P != 0;
    if (accumulate) {
        // the flat buffer of the batch's first backward: its first section is the dL_dmeans3D the caller kept
        const Tensor& first = *accumulate_into;
        TORCH_CHECK(!sh_factored, "accumulate_into: not with the factored dL/dSH");
        TORCH_CHECK(first.is_cuda() && first.device() == dev && first.scalar_type() == torch::kFloat32 && first.storage_offset() == 0 &&
                        (long long)(first.storage().nbytes() / sizeof(float)) >= total,
                    "accumulate_into must be the dL_dmeans3D an earlier backward of the same shapes returned");
        flat = torch::empty({0}, f32).set_(first.storage(), 0, {total}, {1});
        // the pool may still list this buffer with the FIRST view's radii ("rows invisible then hold zeros"): not after this call
        first.unsafeGetTensorImpl()->bump_version();
    } else if (P != 0) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool_on) {
            auto& v = g_pool[key];
            for (size_t i = 0; i < v.size(); i++) {
                if (v[i].all.storage().use_count() != 1 || v[i].all.use_count() != 1) continue;  // still somebody's gradient
                flat = v[i].all;
                if ((int64_t)flat._version() == v[i].version) {
                    prev_radii = v[i].prev_radii;
                    g_pool_hits++;
                } else {
                    g_pool_dirty++;  // written to in place since: every row is written again
                }
                v.erase(v.begin() + i);
                break;
            }
        }
    }
    if (!flat.defined()) {
        flat = torch::empty({total}, f32);
        g_pool_fresh++;
    }
    auto view = [&](int i, at::IntArrayRef shape) { return flat.narrow(0, offs[i], sizes[i]).view(shape); };
    Tensor dL_dmeans3D = view(0, {P, 3});
    Tensor dL_dsh = sh_factored ? Tensor() : view(1, {P, M, 3});
    Tensor dL_dsemantics = view(2, {P, S});
    Tensor dL_dopacity = view(3, {P, 1});
    Tensor dL_dscales = view(4, {P, 3});
    Tensor dL_drotations = view(5, {P, 4});
    Tensor dL_dmeans2D = view(6, {P, 3}), dL_dcolors = view(7, {P, 3});
    Tensor dL_ddepths = view(8, {P, 1}), dL_dconic = view(9, {P, 2, 2});
    Tensor dL_dcov3D = view(10, {P, 6});
    if (P != 0) {
        Arg bg = arg(background, "background", dev), m3 = arg(means3D, "means3D", dev), shs = arg(sh, "sh", dev);
        Arg col = arg(colors, "colors_precomp", dev), sem = arg(semantics, "semantics", dev);
        Arg sca = arg(scales, "scales", dev), rot = arg(rotations, "rotations", dev);
        Arg cov = arg(cov3D_precomp, "cov3D_precomp", dev), vm = arg(viewmatrix, "viewmatrix", dev);
        Arg pm = arg(projmatrix, "projmatrix", dev), cp = arg(campos, "campos", dev), al = arg(alphas, "alphas", dev);
        Arg gc = arg(dL_dout_color, "dL_dout_color", dev), gs = arg(dL_dout_semantic, "dL_dout_semantic", dev);
        Arg gd = arg(dL_dout_depth, "dL_dout_depth", dev), ga = arg(dL_dout_alpha, "dL_dout_alpha", dev);
        TORCH_CHECK_TYPE(radii.scalar_type() == torch::kInt32, "radii must be torch.int32");
        Tensor rad = radii.contiguous();
        GoiRasterScene sc = scene_of(P, degree, sh, S, W, H, bg, m3, shs, col, sem, Arg(), sca, scale_modifier, rot, cov, vm,
                                     pm, cp, tan_fovx, tan_fovy, false, debug);
        void* scratch = backward_scratch(goi_raster_backward_scratch_bytes(scratch_instances > 0 ? scratch_instances : R, S), dev,
                                         stream, means3D.options().dtype(torch::kByte));
        const int r = goi_raster_backward3(
            &sc, R, scratch_instances, accumulate ? GOI_BACKWARD_ACCUMULATE : 0, geomBuffer.data_ptr(), binningBuffer.numel() ? binningBuffer.data_ptr() : nullptr,
            imageBuffer.data_ptr(), rad.data_ptr<int>(), al.p, gc.p, gs.p, gd.p, ga.p, dL_dmeans2D.data_ptr<float>(),
            dL_dconic.data_ptr<float>(), dL_dopacity.data_ptr<float>(), dL_dcolors.data_ptr<float>(),
            dL_dsemantics.data_ptr<float>(), dL_ddepths.data_ptr<float>(), dL_dmeans3D.data_ptr<float>(),
            dL_dcov3D.data_ptr<float>(), dL_dsh.defined() && dL_dsh.numel() ? dL_dsh.data_ptr<float>() : nullptr,
            dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(), scratch,
            prev_radii.defined() ? prev_radii.data_ptr<int>() : nullptr, stream);
        if (r < 0) raise_last();
        // the buffer goes (back) into the pool with this frame's radii; it is handed out again only when every view the
        // caller got has been released and nothing has written to it in place
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool_on && !debug && !accumulate) {  // (an accumulated buffer is the caller's: its zero rows are no longer this frame's)
            auto& v = g_pool[key];
            if (v.size() >= POOL_BUFFERS_PER_KEY) v.erase(v.begin());
            v.push_back(PoolEntry{flat, rad, (int64_t)flat._version()});
        }
    }
    flat = Tensor();  // (the pool's reference is the only one besides the views returned below)
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
                           dL_dscales, dL_drotations);
}

// the reference's signature (rasterize_points.cu:213-306): all four upstream gradients are tensors
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians_backward(
    const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& colors, const Tensor& semantics,
    const Tensor& scales, const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp,
    const Tensor& viewmatrix, const Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
    const Tensor& dL_dout_color, const Tensor& dL_dout_semantic, const Tensor& dL_dout_depth, const Tensor& dL_dout_alpha,
    const Tensor& sh, const int degree, const Tensor& campos, const Tensor& geomBuffer, const int R,
    const Tensor& binningBuffer, const Tensor& imageBuffer, const Tensor& alphas, const bool debug) {
    return backward_ex(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier, cov3D_precomp,
                       viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_semantic, dL_dout_depth,
                       dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug, false, 0, c10::nullopt);
}

// dL/dsemantics only (goi_raster_backward_semantics)
Tensor backward_semantics(const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& semantics,
                          const Tensor& viewmatrix, const Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                          const Tensor& dL_dout_semantic, const Tensor& campos, const Tensor& geomBuffer, const int R,
                          const Tensor& binningBuffer, const Tensor& imageBuffer, const Tensor& alphas, const int sh_degree,
                          const bool debug) {
    const c10::Device dev = check_device(means3D);
    c10::hip::HIPGuard guard(dev.index());
    const int P = (int)means3D.size(0), S = (int)dL_dout_semantic.size(0);
    const int H = (int)dL_dout_semantic.size(1), W = (int)dL_dout_semantic.size(2);
    Tensor dL_dsemantics = torch::empty({P, S}, means3D.options().dtype(torch::kFloat32));
    if (P != 0) {
        Arg bg = arg(background, "background", dev), m3 = arg(means3D, "means3D", dev), sem = arg(semantics, "semantics", dev);
        Arg vm = arg(viewmatrix, "viewmatrix", dev), pm = arg(projmatrix, "projmatrix", dev), cp = arg(campos, "campos", dev);
        Arg al = arg(alphas, "alphas", dev), gs = arg(dL_dout_semantic, "dL_dout_semantic", dev);
        Tensor rad = radii.contiguous();
        // the geometry inputs are not read by this path; the scene only has to pass validation
        GoiRasterScene sc = scene_of(P, sh_degree, Tensor(), S, W, H, bg, m3, Arg(), m3, sem, Arg(), Arg(), 1.0f, Arg(), m3, vm,
                                     pm, cp, tan_fovx, tan_fovy, false, debug);
        void* stream = stream_of(dev);
        void* scratch = backward_scratch(goi_raster_backward_scratch_bytes(R, S), dev, stream,
                                         means3D.options().dtype(torch::kByte));
        if (goi_raster_backward_semantics(&sc, R, geomBuffer.data_ptr(),
                                          binningBuffer.numel() ? binningBuffer.data_ptr() : nullptr,
                                          imageBuffer.data_ptr(), rad.data_ptr<int>(), al.p, gs.p,
                                          dL_dsemantics.data_ptr<float>(), scratch, stream) < 0)
            raise_last();
    }
    return dL_dsemantics;
}

// ---- trace / mark_visible -------------------------------------------------------------------------------------------
std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians_trace(
    const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& img_sem, const Tensor& opacity,
    const Tensor& scales, const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp,
    const Tensor& viewmatrix, const Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const Tensor& sh, const int degree, const Tensor& campos, const bool prefiltered,
    const bool debug) {
    TORCH_CHECK(img_sem.defined() && img_sem.numel() > 0, "img_sem [S,H,W] is required");
    Prepared f = prepare(background, means3D, colors, img_sem, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                         viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                         prefiltered, debug, false);
    c10::hip::HIPGuard guard(f.dev.index());
    Arg img_in = arg(img_sem, "img_sem", f.dev);
    auto f32 = means3D.options().dtype(torch::kFloat32);
    auto bytes = means3D.options().dtype(torch::kByte);
    Tensor out_color = torch::empty({3, f.H, f.W}, f32);
    Tensor gau_sem = torch::zeros({f.P, f.S}, f32);
    Tensor num_gsem = torch::zeros({f.P}, means3D.options().dtype(torch::kInt32));
    Tensor radii = torch::empty({f.P}, means3D.options().dtype(torch::kInt32));
    Tensor geom = torch::empty({f.P > 0 ? (long long)goi_raster_geom_bytes(f.P) : 0}, bytes);
    Tensor img = torch::empty({f.P > 0 ? (long long)goi_raster_image_bytes(f.W, f.H) : 0}, bytes);
    Tensor binning = torch::empty({0}, bytes);
    const int n = goi_raster_trace(&f.sc, img_in.p, f.P ? geom.data_ptr() : nullptr, f.P ? img.data_ptr() : nullptr, grow,
                                   &binning, out_color.data_ptr<float>(), f.P ? gau_sem.data_ptr<float>() : nullptr,
                                   f.P ? num_gsem.data_ptr<int>() : nullptr, f.P ? radii.data_ptr<int>() : nullptr,
                                   stream_of(f.dev));
    if (n < 0) raise_last();
    return std::make_tuple(n, out_color, gau_sem, num_gsem, geom, binning, img);
}

Tensor mark_visible(const Tensor& means3D, const Tensor& viewmatrix, const Tensor& projmatrix) {
    const c10::Device dev = check_device(means3D);
    c10::hip::HIPGuard guard(dev.index());
    const int P = (int)means3D.size(0);
    Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        Arg m = arg(means3D, "means3D", dev), v = arg(viewmatrix, "viewmatrix", dev), p = arg(projmatrix, "projmatrix", dev);
        if (goi_raster_mark_visible(P, m.p, v.p, p.p, reinterpret_cast<uint8_t*>(present.data_ptr<bool>()), stream_of(dev)) < 0)
            raise_last();
    }
    return present;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled torch binding of libgoi_raster.so (reference surface: ext.cpp:15-20)";
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
    m.def("rasterize_gaussians_trace", &rasterize_gaussians_trace);
    m.def("mark_visible", &mark_visible);
    m.def("rasterize_gaussians_async", &rasterize_gaussians_async, "speculative forward", pybind11::arg("background"),
          pybind11::arg("means3D"), pybind11::arg("colors"), pybind11::arg("semantics"), pybind11::arg("opacity"),
          pybind11::arg("scales"), pybind11::arg("rotations"), pybind11::arg("scale_modifier"), pybind11::arg("cov3D_precomp"),
          pybind11::arg("viewmatrix"), pybind11::arg("projmatrix"), pybind11::arg("tan_fovx"), pybind11::arg("tan_fovy"),
          pybind11::arg("image_height"), pybind11::arg("image_width"), pybind11::arg("sh"), pybind11::arg("degree"),
          pybind11::arg("campos"), pybind11::arg("prefiltered"), pybind11::arg("debug"), pybind11::arg("capacity"),
          pybind11::arg("zcut_in") = pybind11::none(), pybind11::arg("zcut_out") = pybind11::none());
    m.def("backward_ex", &backward_ex);
    m.def("backward_semantics", &backward_semantics);
    m.def("release_scratch", &release_scratch);
    m.def("set_grad_pool", &set_grad_pool);
    m.def("grad_pool_stats", &grad_pool_stats);
    // the header this binding was COMPILED against (a stale _goi_C.so next to a newer library must be detectable) ...
    m.def("abi_version", []() { return (int)GOI_RASTER_ABI_VERSION; });
    // ... and what the library it is linked with reports at run time
    m.def("library_abi_version", []() { return goi_raster_abi_version(); });
}
