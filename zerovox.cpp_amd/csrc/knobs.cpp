// knobs.cpp — see knobs.h
#include "knobs.h"

#include <cstdlib>
#include <cstring>

namespace zv
{

namespace
{
struct Entry
{
    const char *name;
    int         value, dflt;
};
#define K(n, d) {#n, d, d}
Entry g_knobs[ZV_KNOB_COUNT] = {
    K(ZV_NO_FUSE, 0),       K(ZV_NO_TRIPLE, 0),   K(ZV_FUSE256, 0),       K(ZV_NO_MERGE, 0),     K(ZV_MERGE_ALWAYS, 0), K(ZV_MERGE_SEQ, 1), K(ZV_MERGE_MAXC, 256), K(ZV_VOC_GROUP, 0),
    K(ZV_TAIL_GROUPS, 4),   K(ZV_ARENA_FILL, 0),  K(ZV_DEC_PREPASS, -1),  K(ZV_DBG, 0),          K(ZV_CONV_MT, 0),
    K(ZV_CONV_NT, 0),       K(ZV_CONV_SINGLE, 1), K(ZV_CONV_GEMM, 1), K(ZV_GEMM_ORDER, 2), K(ZV_CONV_LW, 0), K(ZV_PAIR_MT, 0),       K(ZV_PAIR64_RING, 1),  K(ZV_TRIPLE_CFG, 0),
    K(ZV_TRIPLE_V2, 1),     K(ZV_TRIPLE_DB, 1),   K(ZV_TRIPLE_INTERLEAVE, 1), K(ZV_ATT_SCALAR, 0), K(ZV_ATT_MFMA, 0),
    K(ZV_TAIL_FUSED, 1),    K(ZV_STAMP_CP, 0),    K(ZV_STAMP_CONV, 0),    K(ZV_STAMP_CIN, 0),
};
#undef K

// the one place the environment is read: when the library is loaded
struct Init
{
    Init()
    {
        for (Entry &e : g_knobs)
            if (const char *v = getenv(e.name)) e.value = atoi(v);
    }
} g_init;
}  // namespace

int knob(Knob k) { return g_knobs[k].value; }

const char *knob_name(int k) { return k >= 0 && k < ZV_KNOB_COUNT ? g_knobs[k].name : nullptr; }

void knob_reset()
{
    for (Entry &e : g_knobs) e.value = e.dflt;
}

bool knob_set(const char *name, int value)
{
    for (Entry &e : g_knobs)
        if (strcmp(e.name, name) == 0)
        {
            e.value = value;
            return true;
        }
    return false;
}

}  // namespace zv
