// pipeline/render.hpp -- host side of the render pipeline: flatten the AoS scene to the SoA
// the C-ABI takes, upload, render.  This is what src/main.cu:239-247 (render<<<>>> +
// cudaDeviceSynchronize on managed memory) becomes.  HIP is reached only through mrt.h.
#pragma once
#include <cstdint>
#include <variant>
#include <vector>

#include "container/image_buffer.hpp"
#include "container/vector.hpp"
#include "core/light.hpp"
#include "core/scene_object.hpp"
#include "hip_utils/check_error.hpp"
#include "hip_utils/work_partition.hpp"
#include "mrt.h"
#include "pipeline/viewport.hpp"

namespace rt {

// SoA image of a scene, in the layout of mrt_scene_desc
struct FlatScene {
  std::vector<std::uint8_t> mat_kind;
  std::vector<float> mat_rgb;
  std::vector<float> sphere_cx, sphere_cy, sphere_cz, sphere_r;
  std::vector<std::int32_t> sphere_obj;
  std::vector<float> tri; // 16 streams of num_triangles
  std::vector<std::int32_t> tri_obj;
  std::vector<std::uint8_t> light_kind;
  std::vector<float> light_vec, light_rgb;

  mrt_scene_desc desc() const
  {
    mrt_scene_desc d{};
    d.num_objects = static_cast<std::int32_t>(mat_kind.size());
    d.mat_kind = mat_kind.data();
    d.mat_rgb = mat_rgb.data();
    d.num_spheres = static_cast<std::int32_t>(sphere_obj.size());
    d.sphere_cx = sphere_cx.data(); d.sphere_cy = sphere_cy.data(); d.sphere_cz = sphere_cz.data();
    d.sphere_r = sphere_r.data(); d.sphere_obj = sphere_obj.data();
    d.num_triangles = static_cast<std::int32_t>(tri_obj.size());
    d.tri = tri.data(); d.tri_obj = tri_obj.data();
    d.num_lights = static_cast<std::int32_t>(light_kind.size());
    d.light_kind = light_kind.data(); d.light_vec = light_vec.data(); d.light_rgb = light_rgb.data();
    return d;
  }
};

inline void flatten_lights(FlatScene &f, VectorView<Light> lights)
{
  for (const Light &l : lights) {
    if (const auto *d = std::get_if<DirectionalLight>(&l)) {
      f.light_kind.push_back(MRT_LIGHT_DIRECTIONAL);
      for (int i = 0; i < 3; ++i) { f.light_vec.push_back(d->dir()[i]); f.light_rgb.push_back(d->color()[i]); }
    } else {
      const auto &p = std::get<PointLight>(l);
      f.light_kind.push_back(MRT_LIGHT_POINT);
      for (int i = 0; i < 3; ++i) { f.light_vec.push_back(p.pos()[i]); f.light_rgb.push_back(p.color()[i]); }
    }
  }
}

// Type-partition the object list; every primitive remembers its object number, on which
// closest-hit ties and the ID buffer are defined (src/main.cu:67-77).
inline FlatScene flatten(VectorView<SceneObject> objects, VectorView<Light> lights)
{
  FlatScene f;
  std::vector<float> tri_aos; // 16 floats per triangle, transposed to streams below
  std::int32_t index = 0;
  for (const SceneObject &o : objects) {
    f.mat_kind.push_back(static_cast<std::uint8_t>(o.material().index()));
    const Vec3f col = std::visit([](const auto &m) { return m.color(); }, o.material());
    for (int i = 0; i < 3; ++i) f.mat_rgb.push_back(col[i]);
    if (const auto *s = std::get_if<Sphere>(&o.geometry())) {
      f.sphere_cx.push_back(s->center().x); f.sphere_cy.push_back(s->center().y); f.sphere_cz.push_back(s->center().z);
      f.sphere_r.push_back(s->radius());
      f.sphere_obj.push_back(index);
    } else if (const auto *t = std::get_if<Triangle>(&o.geometry())) {
      float q[16];
      t->precomputed(q);
      tri_aos.insert(tri_aos.end(), q, q + 16);
      f.tri_obj.push_back(index);
    } // Plane never hits (core/plane.hpp): it only keeps its object number
    ++index;
  }
  const std::size_t nt = f.tri_obj.size();
  f.tri.resize(16 * nt);
  for (std::size_t i = 0; i < nt; ++i)
    for (int s = 0; s < 16; ++s) f.tri[s * nt + i] = tri_aos[16 * i + s];
  flatten_lights(f, lights);
  return f;
}

// One GPU.  Fail-fast like the reference's main(): any backend error prints and exits.
class Renderer {
public:
  explicit Renderer(int device = 0) { checkRtErrors(mrt_ctx_create(device, &ctx_)); }
  ~Renderer() { mrt_ctx_destroy(ctx_); }
  Renderer(const Renderer &) = delete;
  Renderer &operator=(const Renderer &) = delete;

  void upload(const FlatScene &f)
  {
    mrt_scene_desc d = f.desc();
    checkRtErrors(mrt_scene_upload(ctx_, &d));
  }
  void upload(VectorView<SceneObject> objects, VectorView<Light> lights) { upload(flatten(objects, lights)); }

  // whole frame on this GPU, blocking
  mrt_stats render(const Viewport &viewport, ImageBuffer &img, int spp = 1, int max_depth = 0)
  {
    mrt_camera cam = viewport.to_abi();
    mrt_render_params p{};
    p.width = static_cast<std::int32_t>(img.width());
    p.height = static_cast<std::int32_t>(img.height());
    p.spp = spp;
    p.max_depth = max_depth;
    p.part = make_strip_partition(0, 1);
    mrt_stats st{};
    checkRtErrors(mrt_render(ctx_, &cam, &p, &img.buffer()[0].x, img.ids(), img.ts(), &st));
    return st;
  }

  // seed of the per-pixel random streams of diffuse bounces; the reference hard-codes 1984
  // (cuda_utils/curand_state.cuh:22), which is also the back end's default
  void seed_random_streams(int seed) { checkRtErrors(mrt_ctx_set_option(ctx_, MRT_OPT_RNG_SEED, seed)); }

  mrt_ctx *ctx() { return ctx_; }

private:
  mrt_ctx *ctx_ = nullptr;
};

// Several GPUs of this node, one process: the scene replicated on every GPU, the frame dealt to them in strips of
// rows (strip s -> GPU s % N), the strips gathered on GPU 0 over RCCL and handed back as one frame
// (mrt_render_multi).  The reference is single GPU (src/main.cu never selects a device); the frame is bit-identical
// to Renderer's.  Fail-fast like Renderer.
class MultiGpuRenderer {
public:
  explicit MultiGpuRenderer(int num_gpus, int strip_rows = 8) : strip_rows_(strip_rows)
  {
    ctxs_.assign(static_cast<std::size_t>(num_gpus > 0 ? num_gpus : 1), nullptr);
    for (std::size_t i = 0; i < ctxs_.size(); ++i) checkRtErrors(mrt_ctx_create(static_cast<int>(i), &ctxs_[i]));
    if (ctxs_.size() > 1) {
      comms_.assign(ctxs_.size(), nullptr);
      checkRtErrors(mrt_comm_create_local(ctxs_.data(), static_cast<std::int32_t>(ctxs_.size()), comms_.data()));
    }
  }
  ~MultiGpuRenderer()
  {
    for (mrt_comm *c : comms_) mrt_comm_destroy(c);
    for (mrt_ctx *c : ctxs_) mrt_ctx_destroy(c);
  }
  MultiGpuRenderer(const MultiGpuRenderer &) = delete;
  MultiGpuRenderer &operator=(const MultiGpuRenderer &) = delete;

  int num_gpus() const { return static_cast<int>(ctxs_.size()); }

  void upload(const FlatScene &f)
  {
    mrt_scene_desc d = f.desc();
    for (mrt_ctx *c : ctxs_) checkRtErrors(mrt_scene_upload(c, &d));
  }
  void upload(VectorView<SceneObject> objects, VectorView<Light> lights) { upload(flatten(objects, lights)); }

  // whole frame, blocking; stats are sums over the GPUs, kernel_ms the slowest GPU's render kernel
  mrt_stats render(const Viewport &viewport, ImageBuffer &img, int spp = 1, int max_depth = 0)
  {
    mrt_camera cam = viewport.to_abi();
    mrt_render_params p{};
    p.width = static_cast<std::int32_t>(img.width());
    p.height = static_cast<std::int32_t>(img.height());
    p.spp = spp;
    p.max_depth = max_depth;
    p.part = make_strip_partition(0, num_gpus(), strip_rows_);
    mrt_stats st{};
    checkRtErrors(mrt_render_multi(ctxs_.data(), comms_.empty() ? nullptr : comms_.data(), num_gpus(), &cam, &p,
                                   &img.buffer()[0].x, img.ids(), img.ts(), &st));
    return st;
  }

  void seed_random_streams(int seed)
  {
    for (mrt_ctx *c : ctxs_) checkRtErrors(mrt_ctx_set_option(c, MRT_OPT_RNG_SEED, seed));
  }

private:
  std::vector<mrt_ctx *> ctxs_;
  std::vector<mrt_comm *> comms_;
  int strip_rows_;
};

} // namespace rt
