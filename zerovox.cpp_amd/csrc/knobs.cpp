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
#define ZV_KNOB_ENTRY(n, d) {#n, d, d},
Entry g_knobs[ZV_KNOB_COUNT] = {ZV_KNOB_LIST(ZV_KNOB_ENTRY) ZV_KNOB_LIST_DIAG(ZV_KNOB_ENTRY)};
unsigned g_epoch = 0;
#undef ZV_KNOB_ENTRY

#ifdef ZV_DIAG
// diagnostic builds: the one place the environment is read, when the library is loaded
struct Init
{
    Init()
    {
        for (Entry &e : g_knobs)
            if (const char *v = getenv(e.name)) e.value = atoi(v);
    }
} g_init;
#endif
}  // namespace

int knob(Knob k) { return g_knobs[k].value; }

const char *knob_name(int k) { return k >= 0 && k < ZV_KNOB_COUNT ? g_knobs[k].name : nullptr; }

void knob_reset()
{
    for (Entry &e : g_knobs) e.value = e.dflt;
    g_epoch++;
}

unsigned knob_epoch() { return g_epoch; }

bool knob_get(const char *name, int *value)
{
    for (Entry &e : g_knobs)
        if (strcmp(e.name, name) == 0)
        {
            *value = e.value;
            return true;
        }
    return false;
}

bool knob_set(const char *name, int value)
{
    for (Entry &e : g_knobs)
        if (strcmp(e.name, name) == 0)
        {
            e.value = value;
            g_epoch++;
            return true;
        }
    return false;
}

}  // namespace zv
