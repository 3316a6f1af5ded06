// Optional per-kernel timing with HIP events recorded on the launch stream (off by default, zero cost then).
// bench.py switches it on to obtain each kernel's average launch duration; the same figures come out of
// `rocprofv3 --kernel-trace --stats` (profiles/).  Two event records per launch cost a few microseconds of stream time
// each, so the timed region of bench.py brackets only the kernel of its roofline line (kamd_profile_select) and the
// full per-kernel table comes from a separate pass.
#pragma once
#include <hip/hip_runtime.h>

namespace kamd {

enum KernelId {
  K_SD_MAIN = 0, K_SD_FINAL, K_SD_GENERIC, K_SD_BACKWARD, K_SDG_BUILD, K_SDG_QUERY,
  K_BIN_FACES, K_RASTER_TILE, K_RASTER_BACKWARD,
  K_SOFT_FILL, K_SOFT_CLASSIFY, K_SOFT_SELECT, K_SOFT_TILE, K_SOFT_BACKWARD, K_SOFT_BACKWARD_LIST,
  K_TD_PREP, K_TD_MAIN, K_TD_FINAL, K_TD_BACKWARD,
  K_VOX_VERTICES, K_VOX_FACES, K_MEMSET, K_PV_FORWARD, K_PV_BACKWARD, K_MESH_INTERSECTION,
  K_DEFTET_FORWARD, K_DEFTET_SORT, K_DEFTET_BACKWARD, K_SPC_STAGE, K_SPC_BUILD, K_MASK_IOU, K_TEXTURE_MAPPING,
  K_WEIGHTED_SUM,
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

}  // namespace kamd
