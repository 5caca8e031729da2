// FFT plans and execution over hipFFT: the counterpart of the reference's ctypes
// binding of cuFFT (reference fft.py:64-202, used by FftTemplate / Fft 205-422).
// hipFFT is loaded on first use (dlopen); nothing on the RFI path needs it.
#include <dlfcn.h>

#include <mutex>

#include "ksp_common.h"

namespace {
typedef void *fftHandle;
struct Fft {
    void *lib = nullptr;
    int (*create)(fftHandle *);
    int (*destroy)(fftHandle);
    int (*set_auto)(fftHandle, int);
    int (*make_plan)(fftHandle, int, long long *, long long *, long long, long long, long long *,
                     long long, long long, int, long long, size_t *);
    int (*set_stream)(fftHandle, hipStream_t);
    int (*set_work)(fftHandle, void *);
    int (*exec_c2c)(fftHandle, void *, void *, int);
    int (*exec_z2z)(fftHandle, void *, void *, int);
    int (*exec_r2c)(fftHandle, void *, void *);
    int (*exec_c2r)(fftHandle, void *, void *);
    int (*exec_d2z)(fftHandle, void *, void *);
    int (*exec_z2d)(fftHandle, void *, void *);
};

const Fft *fft()
{
    static Fft f;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libhipfft.so", "libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"}) {
            f.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (f.lib != nullptr) break;
        }
        if (f.lib == nullptr) return;
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(f.lib, n);
            ok = ok && p != nullptr;
            return p;
        };
        f.create = (decltype(f.create))sym("hipfftCreate");
        f.destroy = (decltype(f.destroy))sym("hipfftDestroy");
        f.set_auto = (decltype(f.set_auto))sym("hipfftSetAutoAllocation");
        f.make_plan = (decltype(f.make_plan))sym("hipfftMakePlanMany64");
        f.set_stream = (decltype(f.set_stream))sym("hipfftSetStream");
        f.set_work = (decltype(f.set_work))sym("hipfftSetWorkArea");
        f.exec_c2c = (decltype(f.exec_c2c))sym("hipfftExecC2C");
        f.exec_z2z = (decltype(f.exec_z2z))sym("hipfftExecZ2Z");
        f.exec_r2c = (decltype(f.exec_r2c))sym("hipfftExecR2C");
        f.exec_c2r = (decltype(f.exec_c2r))sym("hipfftExecC2R");
        f.exec_d2z = (decltype(f.exec_d2z))sym("hipfftExecD2Z");
        f.exec_z2d = (decltype(f.exec_z2d))sym("hipfftExecZ2D");
        if (!ok) {
            dlclose(f.lib);
            f.lib = nullptr;
        }
    });
    return f.lib != nullptr ? &f : nullptr;
}

#define KSP_FFT_CHECK(expr)                                                          \
    do {                                                                             \
        const int _r = (expr);                                                       \
        if (_r != 0) {                                                               \
            ksp_set_error("%s failed: hipfftResult %d (%s:%d)", #expr, _r, __FILE__, \
                          __LINE__);                                                 \
            return (int)hipErrorUnknown;                                             \
        }                                                                            \
    } while (0)
}  // namespace

extern "C" int ksp_fft_plan_create(int device, int rank, const long long *n,
                                   const long long *inembed, long long idist,
                                   const long long *onembed, long long odist, int type,
                                   long long batch, void **plan_out, size_t *work_size)
{
    KSP_REQUIRE(rank >= 1 && rank <= 3, "rank must be 1, 2 or 3");
    KSP_REQUIRE(n != nullptr && inembed != nullptr && onembed != nullptr && plan_out != nullptr &&
                    work_size != nullptr,
                "NULL argument");
    KSP_REQUIRE(type == KSP_FFT_R2C || type == KSP_FFT_C2R || type == KSP_FFT_C2C ||
                    type == KSP_FFT_D2Z || type == KSP_FFT_Z2D || type == KSP_FFT_Z2Z,
                "bad transform type");
    KSP_REQUIRE(batch >= 1, "batch must be positive");
    const Fft *f = fft();
    if (f == nullptr) {
        ksp_set_error("ksp_fft_plan_create: libhipfft.so could not be loaded");
        return (int)hipErrorSharedObjectInitFailed;
    }
    KSP_CHECK(hipSetDevice(device));
    long long dims[3], in[3], out[3];
    for (int i = 0; i < rank; i++) {
        dims[i] = n[i];
        in[i] = inembed[i];
        out[i] = onembed[i];
    }
    fftHandle plan = nullptr;
    KSP_FFT_CHECK(f->create(&plan));
    int rc = f->set_auto(plan, 0);  // the work area is a slot of the operation
    if (rc == 0)
        rc = f->make_plan(plan, rank, dims, in, 1, idist, out, 1, odist, type, batch, work_size);
    if (rc != 0) {
        f->destroy(plan);
        ksp_set_error("hipfftMakePlanMany64 failed: hipfftResult %d", rc);
        return (int)hipErrorInvalidValue;
    }
    *plan_out = plan;
    return 0;
}

extern "C" int ksp_fft_plan_destroy(int device, void *plan)
{
    if (plan == nullptr) return 0;
    const Fft *f = fft();
    KSP_REQUIRE(f != nullptr, "hipFFT is not loaded");
    KSP_CHECK(hipSetDevice(device));
    KSP_FFT_CHECK(f->destroy(plan));
    return 0;
}

extern "C" int ksp_fft_exec(int device, void *stream, void *plan, int type, void *src, void *dest,
                            void *work_area, int inverse)
{
    KSP_REQUIRE(plan != nullptr && src != nullptr && dest != nullptr, "NULL argument");
    const Fft *f = fft();
    KSP_REQUIRE(f != nullptr, "hipFFT is not loaded");
    KSP_CHECK(hipSetDevice(device));
    KSP_FFT_CHECK(f->set_stream(plan, (hipStream_t)stream));
    if (work_area != nullptr) KSP_FFT_CHECK(f->set_work(plan, work_area));
    const int direction = inverse ? 1 : -1;  // HIPFFT_BACKWARD : HIPFFT_FORWARD
    switch (type) {
    case KSP_FFT_C2C: KSP_FFT_CHECK(f->exec_c2c(plan, src, dest, direction)); break;
    case KSP_FFT_Z2Z: KSP_FFT_CHECK(f->exec_z2z(plan, src, dest, direction)); break;
    case KSP_FFT_R2C: KSP_FFT_CHECK(f->exec_r2c(plan, src, dest)); break;
    case KSP_FFT_C2R: KSP_FFT_CHECK(f->exec_c2r(plan, src, dest)); break;
    case KSP_FFT_D2Z: KSP_FFT_CHECK(f->exec_d2z(plan, src, dest)); break;
    case KSP_FFT_Z2D: KSP_FFT_CHECK(f->exec_z2d(plan, src, dest)); break;
    default:
        ksp_set_error("ksp_fft_exec: bad transform type %d", type);
        return (int)hipErrorInvalidValue;
    }
    return 0;
}
