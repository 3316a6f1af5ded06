// Optional per-kernel timing with HIP events on the launch stream (off by default, zero cost then).
// bench.py switches it on to obtain each kernel's average launch duration; the same figures come out of
// `rocprofv3 --kernel-trace --stats` (profiles/).  Two forms:
//   * ProfScope: two event records around whatever is enqueued inside the scope (several launches, a library call).  The
//     interval holds the command processor's gaps on both sides of a kernel (~2-5 us per launch at C4), and the records cost
//     stream time, so the timed region of bench.py times only the kernel of its roofline line (kamd_profile_select) and the
//     full per-kernel table comes from a separate pass;
//   * KAMD_LAUNCH_TIMED (round 5): a single launch through hipExtLaunchKernelGGL, whose two events take the dispatch's OWN
//     begin and end timestamps -- the interval rocprofv3's kernel trace reports.  The DIB-R kernels use it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace kamd {

enum KernelId {
  K_SD_MAIN = 0, K_SD_FINAL, K_SD_GENERIC, K_SD_BACKWARD, K_SDG_BUILD, K_SDG_QUERY,
  K_BIN_FACES, K_RASTER_TILE, K_RASTER_BACKWARD,
  K_SOFT_FILL, K_SOFT_CLASSIFY, K_SOFT_SELECT, K_SOFT_TILE, K_SOFT_BACKWARD, K_SOFT_BACKWARD_LIST,
  K_TD_PREP, K_TD_MAIN, K_TD_FINAL, K_TD_BACKWARD,
  K_VOX_VERTICES, K_VOX_FACES, K_MEMSET, K_PV_FORWARD, K_PV_BACKWARD, K_MESH_INTERSECTION,
  K_DEFTET_FORWARD, K_DEFTET_SORT, K_DEFTET_BACKWARD, K_SPC_STAGE, K_SPC_BUILD, K_MASK_IOU, K_TEXTURE_MAPPING,
  K_WEIGHTED_SUM, K_SOFT_SELECT_ROUNDS,
  K_NUM
};

bool prof_enabled(int id);
bool prof_all();  // every kernel is being timed: callers keep concurrent launches on one stream so that pairs do not overlap
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);

struct ProfScope {
  int id;
  hipStream_t st;
  bool on;
  ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), on(prof_enabled(id_)) {
    if (on) prof_begin(id, st);
  }
  ~ProfScope() {
    if (on) prof_end(id, st);
  }
};

// the events of one launch (both null when the kernel is not being timed)
void prof_kernel_events(hipEvent_t* start, hipEvent_t* stop);
void prof_kernel_done(int id, hipEvent_t start, hipEvent_t stop);
struct ProfKernel {
  int id;
  hipEvent_t start = nullptr, stop = nullptr;
  bool on;
  explicit ProfKernel(int id_) : id(id_), on(prof_enabled(id_)) {
    if (on) prof_kernel_events(&start, &stop);
    on = on && start != nullptr && stop != nullptr;
  }
  void done() {
    if (on) prof_kernel_done(id, start, stop);
  }
};

}  // namespace kamd

// one kernel launch, timed by its own begin / end timestamps when kernel ID is being profiled
#define KAMD_LAUNCH_TIMED(ID, KERNEL, GRID, BLOCK, SHMEM, ST, ...)                                                \
  do {                                                                                                            \
    kamd::ProfKernel pk_(ID);                                                                                     \
    if (pk_.on)                                                                                                   \
      hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, ST, pk_.start, pk_.stop, 0, __VA_ARGS__);                 \
    else                                                                                                          \
      hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, ST, __VA_ARGS__);                                            \
    pk_.done();                                                                                                   \
  } while (0)
