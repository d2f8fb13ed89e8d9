// ABI bookkeeping entry points of libmrgnas_hip.so (see include/mrgnas.h).
#include "common.hpp"

extern "C" int mrg_abi_version(void) { return MRG_ABI_VERSION; }

// lab / tuning knob: upper bound of the grid of the HBM-streaming kernels (common.hpp: stream_grid_for), 64..4096 blocks
extern "C" int mrg_set_stream_blocks(int blocks) {
  if (blocks < 64 || blocks > 4096) return MRG_E_SHAPE;
  mrg::stream_blocks() = blocks;
  return MRG_OK;
}

extern "C" const char* mrg_target_arch(void) { return "gfx950"; }

extern "C" const char* mrg_error_string(int code) {
  switch (code) {
    case MRG_OK: return "success";
    case MRG_E_NULLPTR: return "mrgnas: a required pointer argument is NULL";
    case MRG_E_SHAPE: return "mrgnas: invalid or unsupported size (need D <= 1024, and D <= 256 when D % 4 != 0)";
    case MRG_E_ENUM: return "mrgnas: unknown op / mode / activation code";
    case MRG_E_WORKSPACE: return "mrgnas: workspace pointer missing";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "mrgnas: unknown error code";
}
