// Per-kernel HIP-event timing (see profile.h) and its C entry points.
#include "profile.h"
#include "../../include/kaolin_amd.h"
#include <atomic>
#include <mutex>
#include <vector>

namespace kamd {
namespace {
const char* const kNames[K_NUM] = {
    "sd_main_f32", "sd_final_f32", "sd_forward_generic", "sd_backward", "sdg_build(bbox + cells + scan + scatter)", "sdg_query",
    "bin_faces_kernel", "raster_tile_kernel", "raster_backward_kernel",
    "fill_regions_kernel", "soft_items_kernel", "soft_select_kernel", "soft_eval_kernel", "soft_mask_backward_kernel", "soft_mask_backward_list_kernel",
    "td_prep_kernel", "td_main_kernel", "td_final_kernel", "td_backward_kernel",
    "vox_clear_extent_kernel", "vox_mark_kernel", "zero_fill", "pv_forward_kernel", "pv_backward_kernel", "mesh_intersection_kernel",
    "deftet_forward(pixel sort + search)", "deftet_sort_interp_kernel", "deftet_backward_kernel",
    "mesh_to_spc_stage(count|emit)", "mesh_to_spc_build(sort + unique + octree + results)", "mask_iou_kernels",
    "texture_mapping_kernel", "weighted_sum2_kernels", "soft_select_rounds_kernel"};
struct Pending {
  int id;
  hipEvent_t start, stop;
};
std::atomic<int> g_on{0};
std::atomic<int> g_only{-1};  // >= 0: only this kernel id is timed
std::mutex g_mu;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;
double g_ms[K_NUM];
long long g_cnt[K_NUM];
thread_local hipEvent_t t_start[K_NUM];

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
void drain_locked() {
  for (auto& p : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      g_ms[p.id] += ms;
      g_cnt[p.id] += 1;
    }
    g_pool.push_back(p.start);
    g_pool.push_back(p.stop);
  }
  g_pending.clear();
}
}  // namespace

bool prof_enabled(int id) {
  if (g_on.load(std::memory_order_relaxed) == 0) return false;
  const int only = g_only.load(std::memory_order_relaxed);
  return only < 0 || only == id;
}
bool prof_all() { return g_on.load(std::memory_order_relaxed) != 0 && g_only.load(std::memory_order_relaxed) < 0; }
void prof_begin(int id, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e = get_event();
  t_start[id] = e;
  if (e) (void)hipEventRecord(e, st);
}
void prof_end(int id, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t s = t_start[id];
  hipEvent_t e = get_event();
  if (!s || !e) return;
  (void)hipEventRecord(e, st);
  g_pending.push_back(Pending{id, s, e});
}
void prof_kernel_events(hipEvent_t* start, hipEvent_t* stop) {
  std::lock_guard<std::mutex> lk(g_mu);
  *start = get_event();
  *stop = get_event();
}
void prof_kernel_done(int id, hipEvent_t start, hipEvent_t stop) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending.push_back(Pending{id, start, stop});
}
}  // namespace kamd

extern "C" {
int kamd_profile_enable(int on) {
  kamd::g_on.store(on ? 1 : 0);
  return 0;
}
int kamd_profile_select(int id) {
  kamd::g_only.store(id >= 0 && id < kamd::K_NUM ? id : -1);
  return 0;
}
int kamd_profile_reset(void) {
  std::lock_guard<std::mutex> lk(kamd::g_mu);
  kamd::drain_locked();
  for (int i = 0; i < kamd::K_NUM; ++i) {
    kamd::g_ms[i] = 0;
    kamd::g_cnt[i] = 0;
  }
  return 0;
}
int kamd_profile_num_kernels(void) { return kamd::K_NUM; }
const char* kamd_profile_kernel_name(int id) { return (id >= 0 && id < kamd::K_NUM) ? kamd::kNames[id] : ""; }
int kamd_profile_read(int id, double* total_ms, int64_t* launches) {
  if (id < 0 || id >= kamd::K_NUM) return 1;
  std::lock_guard<std::mutex> lk(kamd::g_mu);
  kamd::drain_locked();
  *total_ms = kamd::g_ms[id];
  *launches = kamd::g_cnt[id];
  return 0;
}
}  // extern "C"
