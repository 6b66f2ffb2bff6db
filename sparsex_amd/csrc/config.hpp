// config.hpp -- process-wide option table (string -> string) with the
// reference's mnemonics and defaults (src/internals/Runtime.cpp:37-95) plus
// the GPU-specific extensions documented in DESIGN.md.
#pragma once

#include "common.hpp"

#include <map>
#include <string>
#include <vector>

namespace spx {

struct XformSpec {
    int type;                    // concrete type or group
    std::vector<size_t> deltas;  // explicit instantiations, may be empty
};

// Parsed form of "spx.preproc.xform", e.g. "h{1,2},bc4{2}" or "all"
// (reference: src/internals/Encodings.cpp:108-138).
struct XformSeq {
    std::vector<XformSpec> seq;
    bool explicit_deltas = false;
};

class Config {
public:
    static Config &instance();

    // returns false (and warns) when the mnemonic is unknown
    bool set(const std::string &mnemonic, const std::string &value);
    void load_from_env();           // Runtime.cpp:97-149
    void reset_defaults();

    std::string get_str(const std::string &mnemonic) const;
    long get_long(const std::string &mnemonic) const;
    double get_double(const std::string &mnemonic) const;
    bool get_bool(const std::string &mnemonic) const;

    // derived views ----------------------------------------------------
    size_t nr_partitions() const;          // spx.rt.nr_threads
    std::vector<size_t> cpu_affinity() const;
    XformSeq xform() const;                // throws FatalError on bad names

private:
    Config();
    std::map<std::string, std::string> props_;
};

}  // namespace spx
