#include "common.h"
#include "../../include/safepo_hip.h"
namespace spo { thread_local char g_err[512] = {0}; }
extern "C" int spo_abi_version(void) { return SPO_ABI_VERSION; }
extern "C" const char* spo_last_error(void) { return spo::g_err; }
