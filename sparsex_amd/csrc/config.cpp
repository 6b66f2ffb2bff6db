// config.cpp -- option table.  Names/defaults: reference Runtime.cpp:37-95
// (non-NUMA build: heuristic "ratio", full_colind "false").
#include "config.hpp"

#include <cstdlib>
#include <regex>
#include <sstream>

namespace spx {

Config &Config::instance()
{
    static Config c;
    return c;
}

Config::Config() { reset_defaults(); }

void Config::reset_defaults()
{
    props_.clear();
    props_["spx.rt.nr_threads"] = "1";
    props_["spx.rt.cpu_affinity"] = "0";
    props_["spx.preproc.heuristic"] = "ratio";
    props_["spx.preproc.xform"] = "all";
    props_["spx.preproc.sampling"] = "portion";
    props_["spx.preproc.sampling.nr_samples"] = "48";
    props_["spx.preproc.sampling.portion"] = "0.01";
    props_["spx.preproc.sampling.window_size"] = "0";
    props_["spx.matrix.symmetric"] = "false";
    props_["spx.matrix.split_blocks"] = "true";
    props_["spx.matrix.full_colind"] = "false";
    props_["spx.matrix.min_unit_size"] = "4";
    props_["spx.matrix.max_unit_size"] = "255";
    props_["spx.matrix.min_coverage"] = "0.1";
    // no mnemonic in the reference (Runtime.cpp:60); kept settable here
    props_["spx.matrix.onedim_blocks"] = "false";
    // ---- extensions of this build (DESIGN.md "Options") ----------------
    props_["spx.rt.host_only"] = "false";    // tune without uploading to a GPU
    props_["spx.rt.gpu_rank"] = "0";         // this process' slice of the partitions
    props_["spx.rt.gpu_world"] = "1";
    props_["spx.rt.device"] = "-1";          // HIP device ordinal, -1 = current
    props_["spx.vec.register"] = "auto";     // auto: a view of a client's buffer (spx_vec_create_from_buff) of 32 MB or more is page-locked in place at its first product; false: always staged
    props_["spx.vec.device"] = "false";      // true: vectors the library creates keep their HBM copy between calls (a client that writes through v->elements must then say so: spx_hip_vec_touch)
    props_["spx.rt.host_parts"] = "0";       // spx_matvec_* on host vectors of 32 MB or more: parts the product runs in while y travels back (and x up) part by part; 0: 24 (symmetric streams: 16) where x goes up by need, else 8; at most 64
    props_["spx.rt.dist_chunks"] = "4";      // SPX_DIST_OVERLAP: parts of the own product the halo exchange is pipelined over (1: no plan)
    props_["spx.rt.dist_reorder"] = "none";  // with gpu_world > 1 and the whole matrix: none | rcm | rcm_owner in front of the cut
    props_["spx.rt.row_offset"] = "0";       // the input holds rows [offset, offset + its rows) ...
    props_["spx.rt.global_rows"] = "0";      // ... of a matrix with this many rows (0: the input is the matrix)
    props_["spx.rt.keep_encoded"] = "true";  // keep the encoded partitions for export
    props_["spx.gpu.rowblock_elems"] = "0";    // target value elements per row-block
    props_["spx.gpu.rowblock_rows"] = "512";   // max rows owned by one row-block
    props_["spx.gpu.stack_segments"] = "true"; // merge stacked row segments into block descriptors
    props_["spx.gpu.waves"] = "0";             // wavefronts per workgroup: 2, 4, 8; 0 = measured at tune time
    props_["spx.gpu.inline_desc"] = "true";    // single-descriptor unit passes carry their descriptor in the pass header
    props_["spx.gpu.arena"] = "false";         // true: one HBM allocation for all arrays of a tuned matrix (measured: no gain, profiles/r04/spread.md)
    props_["spx.gpu.band_order"] = "false";    // launch order: strips across the planes of a stencil (measured slower: off)
    props_["spx.gpu.col_phases"] = "auto";     // general path: column slices launched in turn: 1 (off), 2..8, auto (measured)
    props_["spx.gpu.keep_units"] = "true";     // re-cut: mined units without row neighbours stay units
    props_["spx.gpu.recut_linear"] = "true";   // row segments from one-nonzero-per-lane units where they line up
    props_["spx.gpu.sym_once"] = "true";       // symmetric: dense 8x8 tiles read once, used twice
    props_["spx.gpu.deterministic"] = "false"; // bit-identical repeated products (per-wavefront y tiles, fixed-order hand-overs)
    props_["spx.gpu.wave_tiles"] = "auto";     // a y tile per wavefront instead of one per workgroup: true | false | auto (measured)
    props_["spx.gpu.x_window"] = "true";       // leftovers with nearby columns gather from an LDS window of x
    props_["spx.gpu.sym_segments"] = "auto";   // symmetric: row segments of the lower triangle read once (true | false | auto)
    props_["spx.gpu.sym_segment_min"] = "2";   // shortest run of consecutive columns that is read once
    props_["spx.gpu.sym_segment_max"] = "8";   // ... and the widest segment a longer run is cut into (4: every segment can ride the read-once pipeline)
    props_["spx.gpu.sym_wide_rows"] = "1024";  // rows of a row-block with read-once segments (several planned row-blocks side by side)
    props_["spx.gpu.sym_spill"] = "auto";      // symmetric tiles' transposed sums: lists | atomic | auto (measured)
    props_["spx.gpu.sym_remine"] = "true";     // symmetric: re-cut the mirrored triangle into row segments
    props_["spx.gpu.sym_pure_passes"] = "true"; // symmetric, read-once segments: long runs fill passes of their own (one descriptor, in the header)
    props_["spx.gpu.sym_pipeline"] = "auto";   // ... and those passes run pipelined, x requested with the values (csx_spmv_sx_kernel): true | false | auto (measured)
    props_["spx.gpu.unit_windows"] = "auto";   // general path: the columns of a row-block's unit passes staged in LDS, unit passes pipelined: true | false | auto (measured)
    props_["spx.gpu.unit_window_doubles"] = "3072";  // ... most doubles of x a row-block may stage for them
    props_["spx.gpu.unit_window_gap"] = "16";  // ... column intervals closer than this are staged as one
}

bool Config::set(const std::string &key, const std::string &value)
{
    auto it = props_.find(key);
    if (it == props_.end()) {
        log_msg(LOG_WARN, "mnemonic \"%s\" not found\n", key.c_str());
        return false;
    }
    if (key == "spx.preproc.heuristic" && value != "ratio" && value != "cost") {
        log_msg(LOG_ERR, "invalid value \"%s\" while setting property "
                "\"spx.preproc.heuristic\"\n", value.c_str());
        throw FatalError("invalid heuristic");
    }
    if (key == "spx.preproc.sampling" && value != "none" && value != "window" &&
        value != "portion") {
        log_msg(LOG_ERR, "invalid value \"%s\" while setting property "
                "\"spx.preproc.sampling\"\n", value.c_str());
        throw FatalError("invalid sampling method");
    }
    it->second = value;
    return true;
}

void Config::load_from_env()
{
    const char *s;
    if ((s = getenv("SYMMETRIC"))) set("spx.matrix.symmetric", s);
    if ((s = getenv("CPU_AFFINITY"))) set("spx.rt.cpu_affinity", s);
    if ((s = getenv("NUM_THREADS"))) set("spx.rt.nr_threads", s);
    if ((s = getenv("XFORM_CONF"))) set("spx.preproc.xform", s);
    if ((s = getenv("WINDOW_SIZE"))) {
        set("spx.preproc.sampling", "window");
        set("spx.preproc.sampling.window_size", s);
    }
    if ((s = getenv("SAMPLES"))) set("spx.preproc.sampling.nr_samples", s);
    if ((s = getenv("SAMPLING_PORTION"))) {
        set("spx.preproc.sampling", "portion");
        set("spx.preproc.sampling.portion", s);
    }
    if ((s = getenv("SAMPLING"))) set("spx.preproc.sampling", s);
    // (this build's: lets an unchanged client binary that calls spx_options_set_from_env opt in to resident vectors)
    if ((s = getenv("SPX_VEC_DEVICE"))) set("spx.vec.device", s);
}

std::string Config::get_str(const std::string &key) const
{
    auto it = props_.find(key);
    if (it == props_.end()) {
        log_msg(LOG_ERR, "property \"%s\" not found\n", key.c_str());
        throw FatalError("unknown property");
    }
    return it->second;
}

long Config::get_long(const std::string &key) const
{
    std::string v = get_str(key);
    char *end = nullptr;
    long r = strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end != '\0') {
        log_msg(LOG_ERR, "invalid value \"%s\" while setting property \"%s\"\n",
                v.c_str(), key.c_str());
        throw FatalError("bad integer property");
    }
    return r;
}

double Config::get_double(const std::string &key) const
{
    std::string v = get_str(key);
    char *end = nullptr;
    double r = strtod(v.c_str(), &end);
    if (end == v.c_str() || *end != '\0') {
        log_msg(LOG_ERR, "invalid value \"%s\" while setting property \"%s\"\n",
                v.c_str(), key.c_str());
        throw FatalError("bad floating-point property");
    }
    return r;
}

bool Config::get_bool(const std::string &key) const
{
    // `true'/`false' only, like the boolalpha extraction in Runtime.hpp:158-166
    return get_str(key) == "true";
}

size_t Config::nr_partitions() const
{
    long n = get_long("spx.rt.nr_threads");
    if (n < 1) n = 1;
    return (size_t) n;
}

std::vector<size_t> Config::cpu_affinity() const
{
    // Runtime.cpp:151-205: comma list; when its length differs from the
    // thread count, cpus 1..T-1 are appended to whatever was parsed.
    std::vector<size_t> aff;
    std::stringstream ss(get_str("spx.rt.cpu_affinity"));
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        if (tok.empty()) continue;
        char *end = nullptr;
        long v = strtol(tok.c_str(), &end, 10);
        if (end == tok.c_str() || *end != '\0' || v < 0) {
            log_msg(LOG_ERR, "invalid value \"%s\" while setting property "
                    "\"spx.rt.cpu_affinity\"\n", tok.c_str());
            throw FatalError("bad affinity");
        }
        aff.push_back((size_t) v);
    }
    size_t nt = nr_partitions();
    if (aff.size() != nt)
        for (size_t i = 1; i < nt; ++i) aff.push_back(i);
    return aff;
}

XformSeq Config::xform() const
{
    XformSeq out;
    const std::string s = get_str("spx.preproc.xform");
    static const std::regex syntax("([a-z]+([0-9]*))(\\{([0-9]+(,[0-9]+)*)\\})?");
    auto it = std::sregex_iterator(s.begin(), s.end(), syntax);
    for (; it != std::sregex_iterator(); ++it) {
        const std::smatch &m = *it;
        std::string name = m[1].str();
        int t = enc_from_short_name(name);
        if (t < 0) {
            log_msg(LOG_ERR, "invalid value \"%s\" while setting property "
                    "\"spx.preproc.xform\"\n", name.c_str());
            throw FatalError("bad xform name");
        }
        XformSpec spec;
        spec.type = t;
        std::stringstream ds(m[4].str());
        std::string tok;
        while (std::getline(ds, tok, ',')) {
            if (tok.empty()) continue;
            out.explicit_deltas = true;
            spec.deltas.push_back((size_t) strtoul(tok.c_str(), nullptr, 10));
        }
        out.seq.push_back(spec);
    }
    return out;
}

}  // namespace spx
