// plugins/correlated.so -- discovery symbols of MI_EXPORT_PLUGIN(CorrelatedSampler, "Independent Sampler")
// (src/samplers/correlated.cpp:194-195; include/mitsuba/core/class.h:206-211).
#include "../../include/dtof.h"
extern "C" {
const char *plugin_name() { return "CorrelatedSampler"; }
const char *plugin_descr() { return "Independent Sampler"; }
const char *plugin_backend() { return dtof_version(); }
}
