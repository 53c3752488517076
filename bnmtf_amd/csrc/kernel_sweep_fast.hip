// K3, 2/4/8-wave blocks: the on-chip half sweep (sweep_chip.inc) for shards of a multi-GPU run, small problems and masks
// with more than kWideMaxSlots slots per lane (up to kFastMaxSlots: two waves per SIMD, 256 VGPRs).  8 waves is the
// throughput shape; when a rank owns few units, 4- or 2-wave blocks keep every CU busy instead of a few, and such
// blocks carry one extra wave that does nothing but issue the panels' LDS-DMA (see sweep_chip_body).
#include "sweep_chip.inc"

namespace bnmtf {

bool sweep_fast_supported(int KP, int pw) { return pw <= kChipPanelStride && sweep_chip_lds_bytes(KP, pw, 8) <= 160 * 1024; }

void launch_sweep_fast(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  if (f.nw == 2 || f.nw == 4) launch_sweep_small(a, f, st);         // kernel_sweep_small.hip
  else if (chip_split_enabled()) launch_chip<8, 0, 1>(a, f, st);
  else launch_chip<8, 0, 0>(a, f, st);
}

}  // namespace bnmtf
