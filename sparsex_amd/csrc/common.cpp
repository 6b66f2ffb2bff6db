// common.cpp -- encoding name tables and the minimal logger.
#include "common.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace spx {

namespace {
struct NameRow { int type; const char *shortn; const char *fulln; };
// short names: reference src/internals/Encodings.cpp:32-57
const NameRow kNames[] = {
    {ENC_NONE, "none", "Delta"},
    {ENC_H, "h", "Horizontal"},
    {ENC_V, "v", "Vertical"},
    {ENC_D, "d", "Diagonal"},
    {ENC_AD, "ad", "Antidiagonal"},
    {5, "br1", "BlockRow1"}, {6, "br2", "BlockRow2"}, {7, "br3", "BlockRow3"},
    {8, "br4", "BlockRow4"}, {9, "br5", "BlockRow5"}, {10, "br6", "BlockRow6"},
    {11, "br7", "BlockRow7"}, {12, "br8", "BlockRow8"},
    {13, "bc1", "BlockCol1"}, {14, "bc2", "BlockCol2"}, {15, "bc3", "BlockCol3"},
    {16, "bc4", "BlockCol4"}, {17, "bc5", "BlockCol5"}, {18, "bc6", "BlockCol6"},
    {19, "bc7", "BlockCol7"}, {20, "bc8", "BlockCol8"},
    {ENC_GROUP_BR, "br", "BlockRows"},
    {ENC_GROUP_BC, "bc", "BlockCols"},
    {ENC_GROUP_ALL, "all", "all"},
};
const int kNrNames = sizeof(kNames) / sizeof(kNames[0]);

int g_level = LOG_WARN;
FILE *g_file = nullptr;
std::mutex g_log_mutex;
}  // namespace

const char *enc_short_name(int t)
{
    for (int i = 0; i < kNrNames; ++i)
        if (kNames[i].type == t) return kNames[i].shortn;
    return "?";
}

const char *enc_full_name(int t)
{
    for (int i = 0; i < kNrNames; ++i)
        if (kNames[i].type == t) return kNames[i].fulln;
    return "?";
}

int enc_from_short_name(const std::string &s)
{
    for (int i = 0; i < kNrNames; ++i)
        if (s == kNames[i].shortn) return kNames[i].type;
    return -1;
}

void enc_expand(int t, std::vector<int> &out)
{
    switch (t) {
    case ENC_GROUP_BR:
        for (int i = ENC_BR1; i <= ENC_BR8; ++i) out.push_back(i);
        break;
    case ENC_GROUP_BC:
        for (int i = ENC_BC1; i <= ENC_BC8; ++i) out.push_back(i);
        break;
    case ENC_GROUP_ALL:
        for (int i = ENC_NONE; i < ENC_MAX; ++i) out.push_back(i);
        break;
    default:
        out.push_back(t);
    }
}

void log_set_level(int level) { g_level = level; }

void log_set_file(const char *path)
{
    std::lock_guard<std::mutex> lk(g_log_mutex);
    if (g_file) { fclose(g_file); g_file = nullptr; }
    if (path) g_file = fopen(path, "a");
}

void log_msg(int level, const char *fmt, ...)
{
    if (level > g_level) return;
    static const char *tags[] = {"", "[ERROR]: ", "[WARNING]: ", "[INFO]: ",
                                 "[VERBOSE]: ", "[DEBUG]: "};
    std::lock_guard<std::mutex> lk(g_log_mutex);
    FILE *out = g_file ? g_file : stderr;
    fputs(tags[level], out);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(out, fmt, ap);
    va_end(ap);
    fflush(out);
}

}  // namespace spx
