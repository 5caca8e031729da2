// Run-time compilation and generic kernel launch: the counterpart of the reference's
// AbstractContext.compile / AbstractProgram.get_kernel / enqueue_kernel for USER kernels
// (reference abc.py:160-245, 406-432; cuda.py:182-187 hands the source to nvcc, here it
// goes to hiprtc for the device's own architecture). The operations of this package
// that ship as source templates (fill, hreduce) and any downstream kernel written
// against kernels/port.h go through here; the hot-path kernels are compiled ahead of
// time and never do.
//
// hiprtc is loaded on first use (dlopen), so a process that never compiles anything
// does not depend on it.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "ksp_common.h"

namespace {
typedef struct _hiprtcProgram *rtcProgram;
struct Rtc {
    void *lib = nullptr;
    int (*create)(rtcProgram *, const char *, const char *, int, const char **, const char **);
    int (*compile)(rtcProgram, int, const char **);
    int (*log_size)(rtcProgram, size_t *);
    int (*log)(rtcProgram, char *);
    int (*code_size)(rtcProgram, size_t *);
    int (*code)(rtcProgram, char *);
    int (*destroy)(rtcProgram *);
    const char *(*error_string)(int);
};

const Rtc *rtc()
{
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib != nullptr) break;
        }
        if (r.lib == nullptr) return;
        auto sym = [&](const char *n) { return dlsym(r.lib, n); };
        r.create = (decltype(r.create))sym("hiprtcCreateProgram");
        r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
        r.log_size = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
        r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
        r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
        r.code = (decltype(r.code))sym("hiprtcGetCode");
        r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
        r.error_string = (decltype(r.error_string))sym("hiprtcGetErrorString");
        if (!r.create || !r.compile || !r.log_size || !r.log || !r.code_size || !r.code ||
            !r.destroy) {
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    return r.lib != nullptr ? &r : nullptr;
}
}  // namespace

extern "C" int ksp_rtc_compile(int device, const char *source, const char *const *options,
                               int n_options, void **module_out, char *log, size_t log_capacity)
{
    KSP_REQUIRE(source != nullptr && module_out != nullptr, "NULL argument");
    KSP_REQUIRE(n_options >= 0 && (n_options == 0 || options != nullptr), "bad options");
    if (log != nullptr && log_capacity > 0) log[0] = '\0';
    const Rtc *r = rtc();
    if (r == nullptr) {
        ksp_set_error("ksp_rtc_compile: libhiprtc.so could not be loaded");
        return (int)hipErrorSharedObjectInitFailed;
    }
    KSP_CHECK(hipSetDevice(device));
    hipDeviceProp_t props;
    KSP_CHECK(hipGetDeviceProperties(&props, device));
    std::vector<const char *> opts(options, options + n_options);
    const std::string arch = std::string("--offload-arch=") + props.gcnArchName;
    opts.push_back(arch.c_str());
    rtcProgram prog = nullptr;
    int rc = r->create(&prog, source, "katsdpsigproc_amd_rtc.hip", 0, nullptr, nullptr);
    if (rc != 0) {
        ksp_set_error("hiprtcCreateProgram failed: %s", r->error_string ? r->error_string(rc) : "?");
        return (int)hipErrorUnknown;
    }
    rc = r->compile(prog, (int)opts.size(), opts.data());
    size_t n_log = 0;
    if (r->log_size(prog, &n_log) == 0 && n_log > 1 && log != nullptr && log_capacity > 0) {
        std::string text(n_log, '\0');
        r->log(prog, &text[0]);
        const size_t n = text.size() < log_capacity - 1 ? text.size() : log_capacity - 1;
        memcpy(log, text.data(), n);
        log[n] = '\0';
    }
    if (rc != 0) {
        ksp_set_error("hiprtcCompileProgram failed: %s (see the log)",
                      r->error_string ? r->error_string(rc) : "?");
        r->destroy(&prog);
        return (int)hipErrorInvalidSource;
    }
    size_t n_code = 0;
    std::vector<char> code;
    if (r->code_size(prog, &n_code) != 0 || n_code == 0) {
        ksp_set_error("hiprtcGetCodeSize failed");
        r->destroy(&prog);
        return (int)hipErrorUnknown;
    }
    code.resize(n_code);
    rc = r->code(prog, code.data());
    r->destroy(&prog);
    if (rc != 0) {
        ksp_set_error("hiprtcGetCode failed");
        return (int)hipErrorUnknown;
    }
    hipModule_t module = nullptr;
    KSP_CHECK(hipModuleLoadData(&module, code.data()));
    *module_out = (void *)module;
    return 0;
}

extern "C" int ksp_module_get_function(int device, void *module, const char *name,
                                       void **function_out)
{
    KSP_REQUIRE(module != nullptr && name != nullptr && function_out != nullptr, "NULL argument");
    KSP_CHECK(hipSetDevice(device));
    hipFunction_t fn = nullptr;
    KSP_CHECK(hipModuleGetFunction(&fn, (hipModule_t)module, name));
    *function_out = (void *)fn;
    return 0;
}

extern "C" int ksp_module_unload(int device, void *module)
{
    if (module == nullptr) return 0;
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipModuleUnload((hipModule_t)module));
    return 0;
}

extern "C" int ksp_launch_function(int device, void *stream, void *function, const unsigned *grid,
                                   const unsigned *block, unsigned shared_bytes,
                                   void **kernel_params)
{
    KSP_REQUIRE(function != nullptr && grid != nullptr && block != nullptr, "NULL argument");
    KSP_REQUIRE(grid[0] > 0 && grid[1] > 0 && grid[2] > 0, "empty grid");
    KSP_REQUIRE(block[0] > 0 && block[1] > 0 && block[2] > 0 &&
                    (unsigned long long)block[0] * block[1] * block[2] <= 1024,
                "workgroup of 0 or more than 1024 threads");
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipModuleLaunchKernel((hipFunction_t)function, grid[0], grid[1], grid[2], block[0],
                                    block[1], block[2], shared_bytes, (hipStream_t)stream,
                                    kernel_params, nullptr));
    return 0;
}
