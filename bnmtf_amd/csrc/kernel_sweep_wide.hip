// K3, 16-wave blocks: the on-chip half sweep (sweep_chip.inc) as one block per CU -- 32 units, four waves per SIMD,
// <= 128 VGPRs, at most kWideMaxSlots slots per lane.  The shape for large single-GPU problems (>= 192 such blocks).
#include "sweep_chip.inc"

namespace bnmtf {

bool sweep_wide_supported(int KP, int pw) { return pw <= kChipPanelStride && sweep_chip_lds_bytes(KP, pw, 16) <= 160 * 1024; }

// f describes the pairs this launch owns (f.npairs of them, at most kWideMaxSlots slots each), 16 pairs per block.
void launch_sweep_wide(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  if (chip_split_enabled() || a.mode == kSweepVB) launch_chip<16, 0, 1>(a, f, st); else launch_chip<16, 0, 0>(a, f, st); }

}  // namespace bnmtf
