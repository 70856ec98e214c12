// md_env.h -- every environment switch this library reads, in ONE table (engine/engine_core.cpp).
//
// A run that is reported (bench.py) must be reproducible from its command line: the product reads the environment only
// through scema_env(), whose names are listed in the table with what they do, and scema_md_env_overrides() (C ABI,
// include/scema_md.h) returns the ones that are set, so that the JSON line can show them (config.env_overrides: an empty
// list in a clean run).  A name that is not in the table aborts: a new switch cannot be added without being declared.
#pragma once

const char *scema_env(const char *name);   // getenv() for a declared switch
