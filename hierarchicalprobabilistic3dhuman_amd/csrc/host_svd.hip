// Host side of the head's 3x3 SVD (models/poseMF_shapeGaussian_net.py:137: the reference runs torch.svd on the
// CPU, i.e. LAPACK sgesdd).  The column signs gesdd returns are part of the network's function (they reach the
// child joints through U_proper, :126-130), so the SVD stays on the very same LAPACK routine: this file calls the
// sgesdd_ that PyTorch's own libtorch_cpu.so exports (MKL), with the same job ('A'), column-major layout and
// workspace query torch.linalg.svd uses -- results are bit-identical to torch.svd -- but spreads the independent
// matrices of a kinematic level over a small persistent thread pool instead of torch's sequential loop.
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "hps_common.h"
#include "svd3_gesdd.h"

namespace hps {

typedef void (*sgesdd_fn)(const char* jobz, const int* m, const int* n, float* a, const int* lda, float* s, float* u,
                          const int* ldu, float* vt, const int* ldvt, float* work, const int* lwork, int* iwork,
                          int* info);

static std::atomic<sgesdd_fn> g_sgesdd{nullptr};

static sgesdd_fn resolve_sgesdd() {
    sgesdd_fn fn = g_sgesdd.load();
    if (fn) return fn;
    // libtorch_cpu.so is normally loaded RTLD_LOCAL by the Python import, so look it up by soname first
    void* h = dlopen("libtorch_cpu.so", RTLD_LAZY | RTLD_NOLOAD);
    void* sym = h ? dlsym(h, "sgesdd_") : nullptr;
    if (!sym) sym = dlsym(RTLD_DEFAULT, "sgesdd_");
    fn = reinterpret_cast<sgesdd_fn>(sym);
    if (fn) g_sgesdd.store(fn);
    return fn;
}

// [U | S | V] (row-major 3x3, 3, row-major 3x3) of one row-major 3x3 matrix
static int svd3(sgesdd_fn gesdd, const float* f, float* out, float* work, int lwork) {
    const int three = 3;
    int info = 0, iwork[24];
    float a[9], u[9], vt[9], s[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i + 3 * j] = f[i * 3 + j];           // column-major copy of F
    gesdd("A", &three, &three, a, &three, s, u, &three, vt, &three, work, &lwork, iwork, &info);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            out[r * 3 + c] = u[r + 3 * c];                                   // U
            out[12 + r * 3 + c] = vt[c + 3 * r];                             // V = VT^T
        }
    out[9] = s[0]; out[10] = s[1]; out[11] = s[2];
    return info;
}

class SvdPool {
public:
    static SvdPool& get() {
        // intentionally leaked: the detached workers wait on its condition variable for the life of the process,
        // so it must never be destroyed by static destructors at exit
        static SvdPool* p = new SvdPool;
        return *p;
    }
    int run(sgesdd_fn gesdd, const float* f, float* out, int n, int threads) {
        std::lock_guard<std::mutex> call_lock(call_mu_);
        ensure_workers(threads - 1);
        gesdd_ = gesdd; f_ = f; out_ = out; n_ = n;
        next_.store(0);
        err_.store(0);
        const int helpers = std::min<int>(threads - 1, (int)workers_.size());
        {
            std::lock_guard<std::mutex> lk(mu_);
            pending_ = helpers;
            wanted_ = helpers;
            ++epoch_;
        }
        cv_.notify_all();
        work_loop();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        return err_.load();
    }

private:
    static constexpr int CHUNK = 16;
    void work_loop() {
        int lwork = -1, info = 0, three = 3, iw[24];
        float q = 0.f, dummy[9];
        gesdd_("A", &three, &three, dummy, &three, dummy, dummy, &three, dummy, &three, &q, &lwork, iw, &info);   // workspace query
        lwork = std::max(64, (int)q);
        std::vector<float> work(lwork);
        for (;;) {
            const int i0 = next_.fetch_add(CHUNK);
            if (i0 >= n_) break;
            const int i1 = std::min(n_, i0 + CHUNK);
            for (int i = i0; i < i1; ++i)
                if (svd3(gesdd_, f_ + (size_t)i * 9, out_ + (size_t)i * 21, work.data(), lwork) != 0) err_.store(1);
        }
    }
    void ensure_workers(int want) {
        want = std::min(want, 31);
        while ((int)workers_.size() < want) {
            const int id = (int)workers_.size();
            workers_.emplace_back([this, id] { worker(id); });
            workers_.back().detach();
        }
    }
    void worker(int id) {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (id >= wanted_) continue;       // not needed for this call
            }
            work_loop();
            {
                std::lock_guard<std::mutex> lk(mu_);
                --pending_;
            }
            done_cv_.notify_one();
        }
    }
    std::mutex call_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    unsigned long epoch_ = 0;
    int pending_ = 0, wanted_ = 0;
    sgesdd_fn gesdd_ = nullptr;
    const float* f_ = nullptr;
    float* out_ = nullptr;
    int n_ = 0;
    std::atomic<int> next_{0}, err_{0};
};

}  // namespace hps

using namespace hps;

extern "C" int hps_host_bind_lapack(const char* library_path) {
    if (!library_path) return bad_arg("hps_host_bind_lapack: null path");
    void* h = dlopen(library_path, RTLD_LAZY);
    void* sym = h ? dlsym(h, "sgesdd_") : nullptr;
    if (!sym) {
        set_error("hps_host_bind_lapack: no sgesdd_ in %s", library_path);
        return HPS_E_UNSUPPORTED;
    }
    g_sgesdd.store(reinterpret_cast<sgesdd_fn>(sym));
    return HPS_OK;
}

extern "C" int hps_host_svd3_packed(const float* f_host, float* usv_host, int n, int num_threads) {
    if (!f_host || !usv_host) return bad_arg("hps_host_svd3_packed: null pointer");
    if (n <= 0) return HPS_OK;
    sgesdd_fn gesdd = resolve_sgesdd();
    if (!gesdd) {
        set_error("hps_host_svd3_packed: LAPACK sgesdd_ not found in the process (expected from libtorch_cpu.so)");
        return HPS_E_UNSUPPORTED;
    }
    if (num_threads < 1) num_threads = 1;
    num_threads = std::min(num_threads, std::max(1, n / 32));
    if (SvdPool::get().run(gesdd, f_host, usv_host, n, num_threads) != 0) {
        set_error("hps_host_svd3_packed: sgesdd reported failure");
        return HPS_E_BADARG;
    }
    return HPS_OK;
}

// The device SVD's algorithm (svd3_gesdd.h) compiled for the host: lets the sign agreement with LAPACK be measured and
// tested without a GPU.  Single thread; f_host (n,9) -> usv_host (n,21) packed like hps_host_svd3_packed.
extern "C" int hps_host_svd3_emulated(const float* f_host, float* usv_host, int n, int svd_flavor) {
    if (!f_host || !usv_host) return bad_arg("hps_host_svd3_emulated: null pointer");
    if (svd_flavor != HPS_SVD_ROUNDING_REFERENCE && svd_flavor != HPS_SVD_ROUNDING_FMA) return bad_arg("hps_host_svd3_emulated: svd_flavor");
    int failed = 0;
    for (int i = 0; i < n; ++i) {
        float U[9], S[3], V[9];
        if (!gesdd3::svd3(svd_flavor, f_host + (size_t)i * 9, U, S, V)) ++failed;
        float* o = usv_host + (size_t)i * 21;
        for (int e = 0; e < 9; ++e) { o[e] = U[e]; o[12 + e] = V[e]; }
        o[9] = S[0]; o[10] = S[1]; o[11] = S[2];
    }
    if (failed) { set_error("hps_host_svd3_emulated: %d matrices did not converge / were not finite", failed); return HPS_E_BADARG; }
    return HPS_OK;
}

// Which rounding flavour of csrc/svd3_gesdd.h reproduces the LAPACK this process is bound to (hps_host_bind_lapack)?  MKL picks
// its kernels by the host CPU: on Intel hosts they fuse multiply-adds (flavour HPS_SVD_ROUNDING_FMA), elsewhere they round like
// reference BLAS (HPS_SVD_ROUNDING_REFERENCE) -- and one time in 10^4 that last bit decides the sign of a singular-vector pair.
// 4 096 fixed pseudo-random matrices I + 0.5 N go through sgesdd_ and through both flavours; returns the flavour whose U, S, V
// are bit-identical on all of them, or -1 when neither is (another LAPACK).  Deterministic, ~2 ms, not cached here.
extern "C" int hps_host_svd_flavor(void) {
    sgesdd_fn gesdd = resolve_sgesdd();
    if (!gesdd) { set_error("hps_host_svd_flavor: LAPACK sgesdd_ not found in the process (expected from libtorch_cpu.so)"); return -1; }
    const int n = 4096;
    int lwork = -1, three = 3, info = 0, iwork[24];
    float query = 0.f, dummy[9];
    gesdd("A", &three, &three, dummy, &three, dummy, dummy, &three, dummy, &three, &query, &lwork, iwork, &info);
    lwork = (int)query > 0 ? (int)query : 256;
    std::vector<float> work((size_t)lwork);
    bool match[2] = {true, true};
    uint32_t st = 12345u;
    auto next_uniform = [&] { st = st * 1664525u + 1013904223u; return (float)(st >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < n && (match[0] || match[1]); ++i) {
        float f[9], ref[21];
        for (int e = 0; e < 9; ++e) {
            float g = -6.0f;                                      // sum of 12 uniforms: close enough to a normal deviate
            for (int k = 0; k < 12; ++k) g += next_uniform();
            f[e] = ((e % 4 == 0) ? 1.0f : 0.0f) + 0.5f * g;
        }
        if (svd3(gesdd, f, ref, work.data(), lwork) != 0) continue;
        for (int fl = 0; fl < 2; ++fl) {
            float U[9], S[3], V[9];
            gesdd3::svd3(fl, f, U, S, V);
            bool same = S[0] == ref[9] && S[1] == ref[10] && S[2] == ref[11];
            for (int e = 0; e < 9 && same; ++e) same = U[e] == ref[e] && V[e] == ref[12 + e];
            if (!same) match[fl] = false;
        }
    }
    if (match[HPS_SVD_ROUNDING_FMA]) return HPS_SVD_ROUNDING_FMA;
    if (match[HPS_SVD_ROUNDING_REFERENCE]) return HPS_SVD_ROUNDING_REFERENCE;
    return -1;
}
