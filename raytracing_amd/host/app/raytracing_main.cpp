// app/raytracing_main.cpp -- the reference's CLI, `raytracing <image-filename>`
// (src/main.cu:159-260), on the MI355X back end: build the hard-coded scene, render on one
// GPU through the C-ABI, flip, write a Radiance .hdr.  With the file name alone it renders what the
// reference binary renders: 2048x1024 (src/main.cu:165), 121 samples per pixel (pipeline/multisample_mesh.cuh),
// bounce chains up to config::MAX_RECURSION_DEPTH = 8 (src/main.cu:26), random streams seeded with 1984
// (cuda_utils/curand_state.cuh:22).  Optional: width height, samples (1 | 121), depth (0..8), GPUs (default 1; with
// more the frame is dealt to them in strips and gathered on GPU 0 over RCCL -- same bytes in the file).
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "io/hdr_writer.hpp"
#include "pipeline/default_scene.hpp"
#include "pipeline/render.hpp"

static void print_help_and_exit(const char *name)
{
  std::printf("usage %s <image-filename> [width height [spp [max-depth [gpus]]]]\n", name);
  std::exit(EXIT_FAILURE);
}

int main(int argc, char **argv)
{
  if (argc != 2 && (argc < 4 || argc > 7)) print_help_and_exit(*argv);
  std::size_t W = 2048, H = 1024; // src/main.cu:165
  int spp = 121, max_depth = 8;
  if (argc >= 4) { W = std::strtoul(argv[2], nullptr, 10); H = std::strtoul(argv[3], nullptr, 10); }
  if (argc >= 5) spp = std::atoi(argv[4]);
  if (argc >= 6) max_depth = std::atoi(argv[5]);
  const int gpus = argc == 7 ? std::atoi(argv[6]) : 1;
  if (W == 0 || H == 0 || gpus < 1) print_help_and_exit(*argv);

  rt::ImageBuffer img(W, H);
  auto viewport = rt::config::make_viewport(float(W) / float(H));
  auto scene = rt::config::make_default_scene();

  rt::MultiGpuRenderer renderer(gpus);
  renderer.upload(rt::make_view(scene.objects), rt::make_view(scene.lights));
  mrt_stats st = renderer.render(viewport, img, spp, max_depth); // first call: also sets up the random streams
  const auto t0 = std::chrono::steady_clock::now();
  st = renderer.render(viewport, img, spp, max_depth);
  const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("rendered %zux%zu spp %d depth %d on %d GPU(s): %llu primary + %llu shadow rays, kernel %.3f ms, frame in host memory "
              "(%s) after %.3f ms\n", W, H, spp, max_depth, gpus, (unsigned long long)st.primary_rays,
              (unsigned long long)st.shadow_rays, st.kernel_ms, img.page_locked() ? "page-locked" : "pageable", wall_ms);

  img.flip_vertical();
  std::printf("writing %s...\n", argv[1]);
  if (!rt::write_hdr(argv[1], img)) { std::fprintf(stderr, "cannot write %s\n", argv[1]); return EXIT_FAILURE; }
  std::printf("done!\n");
  return 0;
}
