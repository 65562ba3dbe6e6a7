// knobs.h — what the environment may and may not change.
//
// OPERATIONAL variables (thread count, message cap, verbosity and phase timings on stderr) are read by the shipped library and
// server.  Variables that change NUMERICS, pick research variants or inject failures exist only in builds made with
// -DTSGO_TESTING — libtsgo_hip_testing.so (what the hook-using tests load, toyslam_amd/build.py) and the CPU twin (oracle/Makefile):
// the shipped libtsgo_hip.so / libtsgo_host.so / graph_optimizer do not even contain their names, so a production process cannot
// change its answers, or declare a solver failure, because of an environment it inherited.
#pragma once
#include <cstdlib>

#ifdef TSGO_TESTING
#define TSGO_RESEARCH_ENV(name) std::getenv(name)
#define TSGO_RESEARCH_INT(name, dflt) (std::getenv(name) ? std::atoi(std::getenv(name)) : (dflt))
#define TSGO_RESEARCH_FLOAT(name, dflt) (std::getenv(name) ? std::atof(std::getenv(name)) : (double)(dflt))
#else
#define TSGO_RESEARCH_ENV(name) (static_cast<const char*>(nullptr))
#define TSGO_RESEARCH_INT(name, dflt) (dflt)
#define TSGO_RESEARCH_FLOAT(name, dflt) ((double)(dflt))
#endif
