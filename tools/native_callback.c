/* tools/native_callback.c — a NATIVE host batch callback for measurements (bench.py aux, tools/bench_components.py): the same synthetic
 * integrand the built-in device functors evaluate (include/t4a_testfunctions.h — workload, not algorithm), behind the
 * t4a_gpu_batch_eval_fn signature (include/t4a_gpu.h:189), i.e. what a Rust `batched_f: Fn(&[MultiIndex]) -> Vec<f64>` looks like from
 * the C side (tensorci2.rs:1513-1524, :1862-1882).  One thread by default (the reference calls `f` point by point on one thread);
 * T4A_CB_THREADS > 1 splits a batch over OpenMP threads (what a rayon-parallel closure would do).
 * Built by __graft_entry__.build(): gcc -O3 -fopenmp -shared -fPIC -o tools/libnative_callback.so tools/native_callback.c */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "../include/t4a_testfunctions.h"

typedef struct {
    int32_t fid, n_acc;
    double params[T4A_FN_MAX_PARAMS];
    const uint64_t* weights; /* [n_acc][total] */
    const uint64_t* offset;  /* [n_sites] into a weight row */
    uint64_t total;
    uint64_t calls, points;  /* statistics */
} t4a_native_fn;

int64_t t4a_native_batch_eval(void* ctx, const uint32_t* idx, size_t n_sites, size_t n_pts, double* out)
{
    t4a_native_fn* f = (t4a_native_fn*)ctx;
    f->calls += 1;
    f->points += n_pts;
    static int threads = 0;
    if (threads == 0) {
        const char* e = getenv("T4A_CB_THREADS");
        threads = e ? atoi(e) : 1;
        if (threads < 1) threads = 1;
    }
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1 && n_pts > 4096)
    for (long long p = 0; p < (long long)n_pts; ++p) {
        uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
        const uint32_t* row = idx + (size_t)p * n_sites; /* (n_sites, n_pts) column-major: point p is contiguous */
        for (int k = 0; k < f->n_acc; ++k) {
            uint64_t a = 0;
            const uint64_t* w = f->weights + (size_t)k * f->total;
            for (size_t s = 0; s < n_sites; ++s) a += w[f->offset[s] + row[s]];
            acc[k] = a;
        }
        out[p] = t4a_fn_value(f->fid, acc, f->params);
    }
    return (int64_t)n_pts;
}
