// K3, 2/4/8-wave blocks: the on-chip half sweep (sweep_chip.inc) for shards of a multi-GPU run, small problems and masks
// with more than kWideMaxSlots slots per lane (up to kFastMaxSlots: two waves per SIMD, 256 VGPRs).  8 waves is the
// throughput shape; when a rank owns few units, 4- or 2-wave blocks keep every CU busy instead of a few, and such
// blocks carry one extra wave that does nothing but issue the panels' LDS-DMA (see sweep_chip_body).
#include "sweep_chip.inc"

namespace bnmtf {

bool sweep_fast_supported(int KP, int pw) { return pw <= kChipPanelStride && sweep_chip_lds_bytes(KP, pw, 8) <= 160 * 1024; }

// (the two-chunk instantiations live in this translation unit only)
// an inner extent of two panels (FastArgs::nch == 2): 8-wave blocks, plain shape
static void launch_chip_two_chunks(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const int nx = a.KP / 32;
  if (a.mode == kSweepDraw) { if (nx == 1) launch_chip_inst<1, kSweepDraw, 8, 0, 0, 2>(a, f, st); else launch_chip_inst<2, kSweepDraw, 8, 0, 0, 2>(a, f, st); }
  else                      { if (nx == 1) launch_chip_inst<1, kSweepMode, 8, 0, 0, 2>(a, f, st); else launch_chip_inst<2, kSweepMode, 8, 0, 0, 2>(a, f, st); }
}

bool sweep_two_chunks_plan(int KP, int m, int* mh, int* pw, int* pw1) {
  const int mz = (m + 31) / 32 * 32;
  const int cap = (kChipPanelStride - 32) / 256 * 256;                 // chunk 0 is exactly mh elements + its 32 zero words behind it in LDS
  int h = ((m + 1) / 2 + 255) / 256 * 256;
  if (h > cap) h = cap;
  const int p1 = (mz - h + 32 + 255) / 256 * 256;
  if (h <= 0 || mz - h <= 0 || p1 > kChipPanelStride) return false;
  const int p = std::max((h + 32 + 255) / 256 * 256, p1);
  if (sweep_chip_lds_bytes(KP, p, 8) > 160 * 1024) return false;
  *mh = h; *pw = p; *pw1 = p1;
  return true;
}

#ifdef BNMTF_EXPERIMENTS
// the twin shape (FastArgs::twin): 8 waves, split sampler, two blocks per CU when q is handed over
static void launch_chip_twin(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const int nx = a.KP / 32;
  if (a.mode == kSweepDraw) { if (nx == 1) launch_chip_inst<1, kSweepDraw, 8, 0, 1, 1, 1>(a, f, st); else launch_chip_inst<2, kSweepDraw, 8, 0, 1, 1, 1>(a, f, st); }
  else                      { if (nx == 1) launch_chip_inst<1, kSweepMode, 8, 0, 1, 1, 1>(a, f, st); else launch_chip_inst<2, kSweepMode, 8, 0, 1, 1, 1>(a, f, st); }
}

#endif

void launch_sweep_fast(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
#ifdef BNMTF_EXPERIMENTS
  if (f.twin) { launch_chip_twin(a, f, st); return; }
#endif
  if (f.nch == 2) launch_chip_two_chunks(a, f, st);                 // an inner extent of two LDS panels (9 185 .. 18 368)
  else if (f.nw == 2 || f.nw == 4) launch_sweep_small(a, f, st);    // kernel_sweep_small.hip
  else if (chip_split_enabled() || a.mode == kSweepVB) launch_chip<8, 0, 1>(a, f, st);      // (the variational sweep: split-sampler shapes only)
  else launch_chip<8, 0, 0>(a, f, st);
}

}  // namespace bnmtf
