// Arguments and layout of the device-resident command ring (rhs_ring.h, capi.hip).  Its own
// header: launch.h only forward-declares RingArgs, so work on the ring recompiles the six
// mfma_ring units and capi.hip, not the library.
#pragma once

namespace ddd {

// ---- device-resident command ring (round 6) ------------------------------------------------
// A caller that owns the Runge-Kutta loop (integrate.py:143-169: the driver calls the
// right-hand side once per stage) inside ddd_stream_fork .. ddd_stream_join: ONE persistent
// kernel (rhs_ring.h: substep_ring_kernel) keeps the conv weights in registers and takes
// every ddd_rk_substep call as a 128-byte command from a page-locked ring the host writes --
// no launch, no drain, no 29 KB of weights per wavefront and substep.  Wavefronts never meet:
// each one walks its own row groups through the command sequence.
//
// Two rings.  The HOST ring is page-locked memory the host writes; reading it costs a PCIe
// round trip, and PCIe serves a few hundred million read requests a second: 2 048 wavefronts
// fetching every command themselves (6 requests each) took 40 us per command (measured,
// profiles/r6_ablation.txt).  So the commands are RELAYED into a ring in device memory that
// every wavefront polls instead: a wavefront that does not find its command there takes a
// lock, reads the next kRelaySlots slots of the host ring with one 1 KB wave-wide load and
// copies the ones that are posted -- one wavefront at a time crosses PCIe, once per
// kRelaySlots commands when the host runs ahead.
//
// A slot is eight 16-byte chunks; chunk c = (payload[3 c], payload[3 c + 1], payload[3 c + 2],
// tag), written with ONE 16-byte store each (by the host, and by the relaying lane) and read
// with ONE 16-byte load: a chunk is never seen half-written, and a command is accepted when
// all its chunks carry the tag of the command index expected ((uint32)(index + 1); slots
// start zeroed, a stale slot carries the tag of index - kRingSlots).
//   payload dwords: 0-1 t, 2-3 y_in, 4-5 y_base, 6-7 y_out, 8-9 acc_in, 10-11 acc_out,
//                   12 c1, 13 c2, 14 batch (kRingStop: leave the kernel), 15-17 spare
constexpr int kRingSlots = 256;      // commands the host may run ahead (power of two)
constexpr int kRingChunks = 6;       // chunks in use
constexpr int kRingSlotChunks = 8;   // chunks per slot (128 bytes)
constexpr int kRelaySlots = 8;       // slots one relay pass copies (64 lanes x 16 bytes)
constexpr int kRingStop = -1;
struct RingArgs {
  const void* slots;        // page-locked host memory: [kRingSlots][kRingSlotChunks] x 16 bytes
  void* dev_slots;          // device memory (fine-grained), same layout: the relayed commands
  unsigned* count;          // device: [kRingSlots] wavefront groups done with the slot's command;
                            // [kRingSlots]: the relay lock
  unsigned* done;           // page-locked host: [kRingSlots] tag of the last command every group
                            // of the launch finished in this slot (the host's back-pressure)
  unsigned* status;         // page-locked host: [0] != 0: the watchdog below expired
  unsigned first_index;     // index (mod 2^32) of the first command this launch takes
  unsigned watchdog_ticks;  // s_memrealtime ticks (100 MHz) a group waits for ONE command
                            // before it gives up (a host that died: never in a live process,
                            // whose park thread posts kRingStop after a few idle milliseconds)
};

}  // namespace ddd
