// hmpc_jit.h -- kernels compiled when a problem is created (host code only): the kernel of each wave count WITH THE PROBLEM'S
// SIZES as constants (hmpc_jit_shape::sized; hmpc_jit_prepare_sized in hmpc_capi.hip), and -- the first form of
// round 4, now the second choice -- register kernels per SHAPE for any admissible MLD system:
//
// The reference accepts any MLDSystem at one speed (warm_start_hmpc/controller.py:58-117).  Here the fast kernel --
// hmpc_qp_kernel<NX, NU, NUB, KF, KB, KT, NW> with the static row map: rows in registers, recursions in registers of wave
// 0 -- is a compile-time instantiation; the library ships the two cart-pole shapes of the reference.  For every other
// shape that meets the static row map's requirements (nx + nu <= 16; every [F G] row with at most two input coefficients;
// columns of at most 16 entries; at most 64 Gram entries with terms; at least one binary -- DevProb::static_rows,
// hmpc_pick_kernel) hmpc_create compiles the instantiation FROM THE SAME SOURCE with the toolchain the library was built
// with: one small translation unit per number of waves per node,
//
//     #define HMPC_KERNEL_ONLY
//     #define HMPC_JIT_KC <longest column, even>
//     #include "hmpc_kernel.hip"
//     HMPC_INSTANCE(NX, NU, NUB, KF, KB, KT, NW)
//     extern "C" void hmpc_jit_kernels(void **cold, void **warm) { ... }
//
// built by hipcc into a shared object in an on-disk cache (HMPC_JIT_CACHE, default <library directory>/jit_cache, else
// ~/.cache/hmpc_amd), keyed by the shape and a hash of the kernel sources, and loaded with dlopen: the kernel's host stub
// registers with the HIP runtime like any other translation unit's and is launched through the same function-pointer
// type.  First use of a shape costs one compilation (~20 s, the three wave counts in parallel), later uses a dlopen.
// Without a compiler at run time (HMPC_HIPCC, default /opt/rocm/bin/hipcc), without the sources next to the library, or
// with HMPC_JIT=0, the run-time-sized kernel serves the shape as before (2.7x slower on the cart-pole system).
// hipRTC was considered: it has no C++ standard headers (<type_traits>, <math.h> of the kernel source) and needs the module
// launch API; the offline compiler needs neither and is the compiler the shipped kernels were built with.
#ifndef HMPC_JIT_H
#define HMPC_JIT_H

#include <dlfcn.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

extern char **environ;

struct hmpc_jit_shape {
    int nx, nu, nub, kf, kb, kt, nw, kc;
    // (Round 4 also built the one-wave kernel for TWO waves per SIMD -- 256 registers per lane, amdgpu_waves_per_eu(2, 2) -- where LDS
    // held six or more nodes per CU: 657 k against 547 k QP/s on a random MLD nx = 6, nu = 2 + 3, N = 12.  Withdrawn in round 5: at 256
    // registers this kernel spills hundreds of bytes per lane, and THAT build came out of the compiler wrong, silently -- right
    // statuses and iteration counts, the rays of 9 % of the infeasible nodes scaled by ~1e-43 (an entry nothing reads grown huge) --,
    // right at -O1, with every other wave count, and in the 512-register build of the same source (profiles/r05_compiler_bisect.md).)
    // SIZED kernel (round 4): the run-time-sized kernel (nx = 0) or its streaming form (nx = -1) compiled with the integer
    // sizes of ONE problem as constants -- `sized` is the body of the HMPC_SIZED(p) macro, assignments to the fields of DevProb
    // (hmpc_sized_fields in hmpc_capi.hip); nu = -1 marks the instantiation (its symbols differ from the shipped kernels').
    // The same code paths with immediates for strides, trip counts and divisions: configs[4] 1.4x.
    std::string sized;
    // compiled with the compiler's ILP schedule (-amdgpu-sched-strategy=iterative-ilp): only for binaries listed in the cache's
    // VALIDATED manifest, see sched_flags
    int ilp = 0;
};

namespace hmpc_jit {

inline std::string dir_of_library()
{
    Dl_info info;
    static int anchor;
    if (!dladdr((void *)&anchor, &info) || !info.dli_fname) return ".";
    std::string p = info.dli_fname;
    const size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}

inline bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// A cache directory is code that gets loaded: it must be a directory of THIS user that nobody else can write to.
inline bool writable_dir(const std::string &d)
{
    if (d.empty()) return false;
    (void)mkdir(d.c_str(), 0755);
    struct stat st;
    if (stat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    if (st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return false;
    return access(d.c_str(), W_OK | X_OK) == 0;
}

inline std::string source_dir()
{
    if (const char *e = getenv("HMPC_JIT_SOURCES")) return e;
    return dir_of_library() + "/csrc";
}

inline std::string include_dir()
{
    if (const char *e = getenv("HMPC_JIT_INCLUDE")) return e;
    return dir_of_library() + "/../include";
}

inline std::string cache_dir()
{
    if (const char *e = getenv("HMPC_JIT_CACHE")) { if (writable_dir(e)) return e; }
    std::string d = dir_of_library() + "/jit_cache";
    if (writable_dir(d)) return d;
    if (const char *h = getenv("HOME")) {
        (void)mkdir((std::string(h) + "/.cache").c_str(), 0755);
        d = std::string(h) + "/.cache/hmpc_amd";
        if (writable_dir(d)) return d;
    }
    d = "/tmp/hmpc_amd_jit_" + std::to_string((unsigned long)geteuid()); // (per user; refused above unless it is this user's alone)
    return writable_dir(d) ? d : std::string();
}

// extra compiler flags of the kernels compiled per shape (part of the cache key), e.g. an occupancy attribute:
//   HMPC_JIT_FLAGS='-DHMPC_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(2,2)))'
// A diagnostic build of the library (make check / make stamps) compiles its kernels the same way.
inline std::string extra_flags()
{
    const char *e = getenv("HMPC_JIT_FLAGS");
    std::string f = e ? e : "";
#ifdef HMPC_CHECK
    f += " -DHMPC_CHECK";
#endif
#ifdef HMPC_STAMPS
    f += " -DHMPC_STAMPS";
#endif
    return f;
}

// Instruction scheduling.  These kernels run ONE wave per SIMD: nothing hides a latency but the schedule itself, and the compiler's
// default strategy schedules for occupancy first.  Its ILP strategy (-amdgpu-sched-strategy=iterative-ilp) is worth 8 - 15 %
// (round 4: headline 489 -> 563 k QP/s, N = 40 118 -> 130 k, configs[4] 30.7 -> 32.1 k) -- AND it is the recipe most of the wrong
// binaries of rounds 4 and 5 were compiled with: an experimental scheduler in front of a register allocator that has to spill
// (profiles/r05_compiler_bisect.md; round 5's sweep of 29 problems x 3 wave counts, profiles/r05_variant_sweep.txt: three
// problems with a wrong kernel under the ILP schedule -- one of them did not come back from a launch at all --, the same kernels
// right under the default schedule).  So:
//   * the DEFAULT recipe of a kernel compiled at run time is the compiler's default schedule;
//   * the ILP schedule is used for a binary only if its NAME -- which covers the problem's sizes, the kernel sources, the flags,
//     the architecture and the compiler's identity -- is listed in the cache's VALIDATED manifest: written by
//     tests/gpu_validate_ilp.py, which runs every such binary (1 / 2 / 4 waves, cold and hand-down instantiation) against the
//     oracle in a process of its own under a watchdog.  The manifest is tracked with the sources; any edit of the kernel, the
//     flags or the toolchain changes the names and with them falls back to the default recipe until the validation has run again.
// The register kernels take DPP broadcasts up to 8 row slots per lane and v_readlane beyond, in both recipes (-DHMPC_DPP_FEW: the
// rule of the shipped kernels).  HMPC_JIT_SCHED=<strategy> | default forces one
// schedule for every kernel (experiments, and the validation run itself); a strategy in HMPC_JIT_FLAGS likewise.
inline std::string sched_flags(const hmpc_jit_shape &s)
{
    // (every register kernel: round 4 gave the kernels with two and four waves per node DPP broadcasts whatever their slot count;
    // round 5's sweep met a two-wave kernel with 13 slots that comes out wrong with them and right with v_readlane, as round 4 had
    // met a one-wave kernel: the shipped kernels' rule for all)
    std::string f = s.nx > 0 ? "-DHMPC_DPP_FEW" : "";
    if (extra_flags().find("amdgpu-sched-strategy") != std::string::npos) return f;
    std::string strat = s.ilp ? "iterative-ilp" : "default";
    if (const char *sc = getenv("HMPC_JIT_SCHED")) strat = sc;
    if (strat != "default") f += std::string(f.empty() ? "" : " ") + "-mllvm -amdgpu-sched-strategy=" + strat;
    return f;
}

// Two late passes of this compiler (ROCm 7.2.0, LLVM's AMDGPU back end) are switched off for every kernel of this library.
// Round 4 met compiled kernels that came out WRONG (nodes ending NUMERICAL), which ones moving with the flags.  Round 5 bisected
// two of them with -opt-bisect-limit (tests/gpu_dev_bisect.py, profiles/r05_compiler_bisect.md): each is right up to ONE pass
// execution and wrong from it on --
//   * the one-wave kernel of a random MLD nx = 3, nu = 3 + 6, N = 12: `stack-slot-coloring` after the allocation of the scalar
//     registers (this kernel spills ~1000 scalars to lanes of vector registers; the pass lets spill slots share a lane);
//   * the four-wave kernel of nx = 8, nu = 5 + 2, N = 12: the post-RA `machine-cp` -- the two binaries differ in ONE
//     instruction, a copy `v_accvgpr_mov_b32 a93, a255` out of an AGPR that holds a spilled VGPR ("Reload Reuse"), inside a
//     divergent region of a loop, which the pass deletes.
// Both passes are clean-ups that must not change what a program computes; without them code size grows by 0.5 %, registers
// and scratch of the headline kernel are unchanged, and every kernel known to come out wrong comes out right.  The nets
// (first-use check, second opinion) stay: two bugs found are not all bugs.  HMPC_JIT_SAFE=0 compiles without these flags.
inline std::string safe_flags()
{
    const char *e = getenv("HMPC_JIT_SAFE");
    return (e && atoi(e) == 0) ? "" : "-mllvm -no-stack-slot-sharing -mllvm -disable-copyprop";
}

inline std::string quoted_flags(const hmpc_jit_shape &s) // (each blank-separated flag in single quotes: attributes carry parentheses)
{
    std::stringstream in(extra_flags() + " " + sched_flags(s) + " " + safe_flags());
    std::string tok, out;
    while (in >> tok) out += "'" + tok + "' ";
    return out;
}

inline std::string compiler() { const char *e = getenv("HMPC_HIPCC"); return e ? e : "/opt/rocm/bin/hipcc"; }

// architecture the kernels are compiled for: the one the library itself was built for (csrc/Makefile passes its ARCH)
#ifndef HMPC_ARCH
#define HMPC_ARCH "gfx950"
#endif
inline std::string arch() { const char *e = getenv("HMPC_JIT_ARCH"); return e ? e : HMPC_ARCH; }

// FNV-1a over everything a kernel is compiled from: the kernel sources, this file (it writes the translation unit and the
// command line), the extra flags, the architecture and the compiler's identity (its resolved path, size and modification
// time: another ROCm under the same path is another compiler) -- an edit or an upgrade of any of them is another cache entry
inline uint64_t source_hash()
{
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&hsh](const std::string &t) { for (char ch : t) { hsh ^= (unsigned char)ch; hsh *= 1099511628211ull; } hsh ^= 0xffu; hsh *= 1099511628211ull; };
    mix(extra_flags());
    mix(arch());
    {
        std::string cc = compiler();
        char real[4096];
        if (realpath(cc.c_str(), real)) cc = real;
        struct stat st;
        char id[96] = "";
        if (stat(cc.c_str(), &st) == 0) snprintf(id, sizeof id, ":%lld:%lld", (long long)st.st_size, (long long)st.st_mtime);
        mix(cc + id);
    }
    for (const std::string &f : {source_dir() + "/hmpc_kernel.hip", source_dir() + "/hmpc_device.h", source_dir() + "/hmpc_jit.h", include_dir() + "/hmpc.h"}) {
        std::ifstream in(f, std::ios::binary);
        char buf[4096];
        while (in) {
            in.read(buf, sizeof buf);
            for (std::streamsize i = 0; i < in.gcount(); i++) { hsh ^= (unsigned char)buf[i]; hsh *= 1099511628211ull; }
        }
    }
    return hsh;
}

inline uint64_t fnv(const std::string &t)
{
    uint64_t hsh = 1469598103934665603ull;
    for (char ch : t) { hsh ^= (unsigned char)ch; hsh *= 1099511628211ull; }
    return hsh;
}

inline std::string name_of(const hmpc_jit_shape &s, uint64_t hsh0)
{
    char b[160];
    const std::string fl = sched_flags(s) + "|" + safe_flags();
    const uint64_t hsh = hsh0 ^ fnv(fl); // (the schedule and the switched-off passes are part of the key)
    if (!s.sized.empty() && s.nx > 0) {
        snprintf(b, sizeof b, "hmpc_s_reg_%d_%d_%d_%d_%d_%d_w%d_kc%d%s_%016llx_%016llx", s.nx, s.nu, s.nub, s.kf, s.kb, s.kt, s.nw, s.kc, "",
                 (unsigned long long)fnv(s.sized), (unsigned long long)hsh);
        return b;
    }
    if (!s.sized.empty()) {
        snprintf(b, sizeof b, "hmpc_s_%s_w%d_r%d_%016llx_%016llx", s.nx < 0 ? "stream" : "generic", s.nw, s.kf, (unsigned long long)fnv(s.sized), (unsigned long long)hsh);
        return b;
    }
    return std::string(); // (every kernel is compiled with a problem's sizes: `sized` is never empty)
}

// Environment of the compiler child: this process's, WITHOUT what a profiler or tool has put there to load itself into
// every process (LD_PRELOAD, the rocprofiler / roctracer / HSA tool variables).  Under `rocprofv3 --pmc` the preloaded
// library initialises the GPU in sh, hipcc, clang ... in turn, and each of them then execs the next: the pattern the GPU
// boxes refuse -- the compilation fails and the profile silently measures the shipped kernel.
inline std::vector<std::string> scrubbed_environment()
{
    static const char *const drop[] = {"LD_PRELOAD=", "ROCP", "ROCPROFILER", "ROCTRACER", "ROCTX", "HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "HIP_TOOLS_LIB="};
    std::vector<std::string> env;
    for (char **e = environ; e && *e; e++) {
        bool keep = true;
        for (const char *d : drop) keep = keep && strncmp(*e, d, strlen(d)) != 0;
        if (keep) env.push_back(*e);
    }
    return env;
}

// Starts the compilation of one shape (returns the child's pid, 0 if the object is in the cache already, -1 on failure).
// Is this binary (by the name of its cache entry) in the VALIDATED manifest of the cache it would be loaded from, or of the
// library's own jit_cache (the manifest the tree ships)?  One name per line, '#' comments.
inline bool validated(const std::string &name)
{
    for (const std::string &dir : {cache_dir(), dir_of_library() + "/jit_cache"}) {
        if (dir.empty()) continue;
        std::ifstream in(dir + "/VALIDATED");
        std::string line;
        while (std::getline(in, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (line == name) return true;
        }
    }
    return false;
}

inline std::string build_tag(unsigned seq) // (temporary files of one build: process id and a per-process counter -- two threads
{                                          //  of one process may compile the same problem at the same time)
    char tag[48];
    snprintf(tag, sizeof tag, ".%d_%u", (int)getpid(), seq);
    return tag;
}
inline unsigned next_build_seq() { static std::atomic<unsigned> n{0}; return n.fetch_add(1); }

inline pid_t start_build(const hmpc_jit_shape &s, const std::string &cache, uint64_t hsh, std::string &err, unsigned seq)
{
    const std::string base = cache + "/" + name_of(s, hsh), so = base + ".so";
    if (exists(so)) return 0;
    if (!exists(source_dir() + "/hmpc_kernel.hip")) { err = "kernel sources not found in " + source_dir(); return -1; }
    if (access(compiler().c_str(), X_OK) != 0) { err = "no compiler at " + compiler(); return -1; }
    const std::string tag = build_tag(seq);
    const std::string src = base + tag + ".hip", tmp = base + tag + ".tmp.so";
    {
        std::ofstream out(src);
        // (the kernel template is renamed in every generated unit: its host stubs must not share their names with the shipped
        // library's -- a process that links the library directly has those in its global scope)
        out << "// generated by hmpc_jit.h\n#define HMPC_KERNEL_ONLY\n#define hmpc_qp_kernel hmpc_qp_kernel_jit\n";
        if (s.nx > 0) out << "#define HMPC_JIT_KC " << s.kc << "\n";
        if (!s.sized.empty()) out << "#define HMPC_SIZED(p) " << s.sized << "\n";
        std::stringstream args;
        if (s.nx > 0) args << s.nx << ", " << s.nu << ", " << s.nub << ", " << s.kf << ", " << s.kb << ", " << s.kt << ", " << s.nw;
        else args << s.nx << ", -1, 0, " << s.kf << ", 0, 0, " << s.nw;
        out << "#include \"hmpc_device.h\"\n#include \"hmpc_kernel.hip\"\nHMPC_INSTANCE(" << args.str() << ")\nextern \"C\" void hmpc_jit_kernels(void **cold, void **warm)\n{\n    *cold = (void *)hmpc_qp_kernel<"
            << args.str() << ", false>;\n    *warm = (void *)hmpc_qp_kernel<" << args.str() << ", true>;\n}\nextern \"C\" const char *hmpc_jit_sized(void) { return \"" << s.sized << "\"; }\n";
        if (!out) { err = "cannot write " + src; return -1; }
    }
    // the compiler runs as a CHILD process (posix_spawn, as Python's subprocess does): nothing of this process is replaced
    const std::string cmd = compiler() + " --offload-arch=" + arch() + " -O3 -fPIC -std=c++17 -ffp-contract=fast -Wno-pass-failed " + quoted_flags(s) + " -I '" + include_dir() + "' -I '" +
                            source_dir() + "' -shared -Wl,-Bsymbolic -o '" + tmp + "' '" + src + "' > '" + base + tag + ".log' 2>&1 && mv '" + tmp + "' '" + so + "'";
    pid_t pid = -1;
    const char *argv[] = {"sh", "-c", cmd.c_str(), nullptr};
    const std::vector<std::string> env = scrubbed_environment();
    std::vector<char *> envp;
    for (const std::string &e : env) envp.push_back(const_cast<char *>(e.c_str()));
    envp.push_back(nullptr);
    if (posix_spawn(&pid, "/bin/sh", nullptr, nullptr, (char *const *)argv, envp.data()) != 0) { err = "cannot start the compiler"; return -1; }
    return pid;
}

inline bool finish_build(pid_t pid, const hmpc_jit_shape &s, const std::string &cache, uint64_t hsh, std::string &err, unsigned seq)
{
    const std::string base = cache + "/" + name_of(s, hsh);
    if (pid > 0) {
        int status = 0;
        pid_t got;
        do got = waitpid(pid, &status, 0); while (got < 0 && errno == EINTR);
        const std::string tag = build_tag(seq);
        if (got < 0) {
            // ECHILD: the host application ignores SIGCHLD (SIG_IGN / SA_NOCLDWAIT: the kernel waits until every child has
            // ended, then reports none) or another thread's wait() has reaped the compiler.  Its exit status is lost: the
            // shared object, moved into place only after a successful compilation, tells.
            if (!exists(base + ".so")) { err = "compilation of " + name_of(s, hsh) + ": the compiler's exit status was lost (" + strerror(errno) + ") and no object was produced"; return false; }
            (void)unlink((base + tag + ".hip").c_str());
            (void)unlink((base + tag + ".log").c_str());
            return true;
        }
        if (!(WIFEXITED(status) && WEXITSTATUS(status) == 0)) {
            std::ifstream log(base + tag + ".log");
            std::stringstream ss;
            ss << log.rdbuf();
            std::string t = ss.str();
            err = "compilation of " + name_of(s, hsh) + " failed: " + (t.size() > 600 ? t.substr(t.size() - 600) : t);
            return false;
        }
        (void)unlink((base + tag + ".hip").c_str());
        (void)unlink((base + tag + ".log").c_str());
    }
    return exists(base + ".so");
}

} // namespace hmpc_jit

// Builds (or finds in the cache) the kernels of `count` shapes, all compilations in flight together.  paths: the shared
// objects, empty where a shape could not be built.  No GPU is touched.
inline bool hmpc_jit_build_all(const hmpc_jit_shape *shapes, int count, std::vector<std::string> &paths, std::string &err)
{
    paths.assign(count, std::string());
    const char *e = getenv("HMPC_JIT");
    if (e && atoi(e) == 0) { err = "HMPC_JIT=0"; return false; }
    const std::string cache = hmpc_jit::cache_dir();
    if (cache.empty()) { err = "no writable cache directory"; return false; }
    const uint64_t hsh = hmpc_jit::source_hash();
    std::vector<pid_t> pid(count, -1);
    std::vector<unsigned> seq(count, 0);
    std::vector<int> same(count, -1); // (two wave counts may be served by one kernel: built once)
    for (int i = 0; i < count; i++) {
        for (int j = 0; j < i && same[i] < 0; j++)
            if (hmpc_jit::name_of(shapes[j], hsh) == hmpc_jit::name_of(shapes[i], hsh)) same[i] = j;
        if (same[i] < 0) { seq[i] = hmpc_jit::next_build_seq(); pid[i] = hmpc_jit::start_build(shapes[i], cache, hsh, err, seq[i]); }
    }
    bool all = true;
    for (int i = 0; i < count; i++) {
        if (same[i] >= 0) { paths[i] = paths[same[i]]; all = all && !paths[i].empty(); continue; }
        if (pid[i] >= 0 && hmpc_jit::finish_build(pid[i], shapes[i], cache, hsh, err, seq[i])) paths[i] = cache + "/" + hmpc_jit::name_of(shapes[i], hsh) + ".so";
        else all = false;
    }
    return all;
}

#endif // HMPC_JIT_H
