// plugins/dopplertofpath.so -- the two extern "C" discovery symbols MI_EXPORT_PLUGIN emits
// (include/mitsuba/core/class.h:206-211; src/integrators/dopplertofpath.cpp:329-330), read by
// PluginManager after dlopen("plugins/<type>.so") (src/core/plugin.cpp:28-44,101-123).
// The implementation lives behind the C ABI of libdtof (include/dtof.h).
#include "../../include/dtof.h"
extern "C" {
const char *plugin_name() { return "DopplerToFPathIntegrator"; }
const char *plugin_descr() { return "Doppler ToF Path Tracer integrator"; }
// convenience forwarder so a host that only dlopen()s the plugin can reach the C ABI
const char *plugin_backend() { return dtof_version(); }
}
