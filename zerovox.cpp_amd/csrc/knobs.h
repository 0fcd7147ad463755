// knobs.h — the library's test / measurement switches in ONE table.
//
// None is needed in production.  Tests force a kernel regime (the regimes give the same bits), measurements A/B a tile
// shape.  The table is filled once, when the library is loaded, from environment variables of the same names (so a
// shell script can still A/B a run), and changed afterwards only through the C-ABI test entry zv_debug_set(name, value):
// no launch path reads the environment.
#pragma once

namespace zv
{

enum Knob : int
{
    ZV_NO_FUSE,            // 1: one launch per conv of a residual block
    ZV_NO_TRIPLE,          // 1: never the whole-block kernels
    ZV_FUSE256,            // 1: fused pair kernel for the 256-channel stage at any length
    ZV_NO_MERGE,           // 1: three branch outputs instead of their sum
    ZV_MERGE_ALWAYS,       // 1: the merged MRF sum at any length (default: only with rounds of workgroups to spare)
    ZV_MERGE_SEQ,          // 0: the wide stages' MRF sum by one three-branch workgroup per tile instead of three single-branch launches
    ZV_MERGE_MAXC,         // widest stage whose last dilation pair stores the merged sum (batches)
    ZV_VOC_GROUP,          // G > 0: the vocoder runs G utterances at a time (experiment)
    ZV_TAIL_GROUPS,        // utterance groups of a batch's last vocoder stage (default 4)
    ZV_ARENA_FILL,         // byte a fresh activation arena is filled with (255: NaN patterns)
    ZV_DEC_PREPASS,        // -1 auto, 0 decoder convs normalise on the fly, 1 f16 operand pass
    ZV_DBG,                // timing-only ablation bits (wrong results)
    ZV_CONV_MT,            // minimum tile height of the generic conv kernel
    ZV_CONV_NT,            // 1 / 2: output tiles per wave of the generic conv kernel
    ZV_CONV_SINGLE,        // 0: never the single-utterance MFMA loop, 2: also for one-chunk convs
    ZV_CONV_GEMM,          // 0 never, 1 batches, 2 always: conv_gemm_kernel for wide convs over an f16 operand tensor
    ZV_GEMM_ORDER,         // conv_gemm_kernel's workgroup order: 0 plain (group fastest), 1 one group per XCD, 2 the 9-tile group first
    ZV_CONV_LW,            // 0 never, 1 batches, 2 always: loader waves + double-buffered tile for multi-chunk convs
    ZV_PAIR_MT,            // 2 / 4: tile height of the pair kernels
    ZV_PAIR64_RING,        // 0 never, 1 batches, 2 always: 64-channel pair kernel with the weights through an LDS ring
    ZV_TRIPLE_CFG,         // MT * 1000 + R of the whole-block kernel
    ZV_TRIPLE_V2,          // 0 never, 1 batches, 2 always: whole-block kernel with its weights in LDS
    ZV_TRIPLE_DB,          // 0: one weight buffer for every branch
    ZV_TRIPLE_INTERLEAVE,  // 0: branches not interleaved per XCD
    ZV_ATT_SCALAR,         // 1: scalar attention kernel
    ZV_ATT_MFMA,           // 1: matrix-core attention kernel whatever the size
    ZV_TAIL_FUSED,         // 0 never, 1 auto: the last vocoder stage as one fused launch (upsample + blocks + output conv)
    ZV_STAMP_CP,           // diagnostic build: channel count of the pair launches that write phase stamps
    ZV_STAMP_CONV,         // diagnostic build: grid.y of the conv launches that write phase stamps
    ZV_STAMP_CIN,          // diagnostic build: their input channels
    ZV_KNOB_COUNT
};

int         knob(Knob k);
// false when no knob has that name
bool        knob_set(const char *name, int value);
// every knob back to its built-in default
void        knob_reset();
const char *knob_name(int k);

}  // namespace zv
