#include "errors.h"
namespace tsgo {
static thread_local std::string g_err;
int set_error(int code, const std::string& text) { g_err = text; return code; }
const char* last_error() { return g_err.c_str(); }
}  // namespace tsgo
extern "C" const char* tsgo_last_error(void) { return tsgo::last_error(); }
