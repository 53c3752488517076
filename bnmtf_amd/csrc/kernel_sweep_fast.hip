// K3, 2/4/8-wave blocks: the on-chip half sweep (sweep_chip.inc) for shards of a multi-GPU run, small problems and masks
// with more than kWideMaxSlots slots per lane (up to kFastMaxSlots: two waves per SIMD, 256 VGPRs).  8 waves is the
// throughput shape; when a rank owns few units, 4- or 2-wave blocks keep every CU busy instead of a few, and such
// blocks carry one extra wave that does nothing but issue the panels' LDS-DMA (see sweep_chip_body).
#include "sweep_chip.inc"

namespace bnmtf {

bool sweep_fast_supported(int KP, int pw) { return pw <= kChipPanelStride && sweep_chip_lds_bytes(KP, pw, 8) <= 160 * 1024; }

void launch_sweep_fast(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  // read per launch (not cached): tests flip them inside one process
  const bool dw = getenv("BNMTF_NO_STAGING_WAVE") == nullptr;
  const bool sp = chip_split_enabled();
  if (f.nw == 2) { if (dw) launch_chip<2, 1, 0>(a, f, st); else launch_chip<2, 0, 0>(a, f, st); }
  else if (f.nw == 4) { if (dw) launch_chip<4, 1, 0>(a, f, st); else launch_chip<4, 0, 0>(a, f, st); }
  else if (sp) launch_chip<8, 0, 1>(a, f, st);
  else launch_chip<8, 0, 0>(a, f, st);
}

}  // namespace bnmtf
