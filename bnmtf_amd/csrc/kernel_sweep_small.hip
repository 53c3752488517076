// K3, 2- and 4-wave blocks (+ one staging wave): the on-chip half sweep (sweep_chip.inc) for the few-units-per-CU case,
// i.e. the shards of a multi-GPU run.  A translation unit of its own so that the instantiations compile in parallel.
#include "sweep_chip.inc"

namespace bnmtf {

void launch_sweep_small(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const bool dw = getenv("BNMTF_NO_STAGING_WAVE") == nullptr;      // read per launch (not cached): tests flip it inside one process
  if (f.nw == 2) { if (dw) launch_chip<2, 1, 0>(a, f, st); else launch_chip<2, 0, 0>(a, f, st); }
  else           { if (dw) launch_chip<4, 1, 0>(a, f, st); else launch_chip<4, 0, 0>(a, f, st); }
}

}  // namespace bnmtf
