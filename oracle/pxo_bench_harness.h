/* pxo_bench_harness.h -- ORACLE / TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Timing harness of bench.py's CPU legs.  Round 3 timed its harness rather than the CPU: a Python ThreadPoolExecutor
 * handing out 0.5 ms tasks, 256 pthreads created and joined per 5 ms pass.  Here:
 *   - the worker threads are created ONCE, pinned round-robin to the CPUs the process may run on, and stay up for the
 *     calibration pass and the timed region (pthread barriers in between, no thread creation inside the timed region);
 *   - every thread builds its own state in `init` ON ITS OWN CPU (first touch: the sample's pages it will read are local
 *     to its NUMA node instead of wherever the Python process happened to allocate them);
 *   - the clock is read inside C, by thread 0, between barrier releases;
 *   - a pass is repeated R times back to back without a barrier (R from the calibration pass) so that the timed region is
 *     at least `min_seconds` long -- every thread gets >= 50 ms of work between two barriers.
 * Header-only (static functions) so that the plain-C oracle and the in-place builds of the reference's C++ both use it. */
#ifndef PXO_BENCH_HARNESS_H_
#define PXO_BENCH_HARNESS_H_
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  /* per-thread state, built by the worker thread itself; may be NULL (then `state` is NULL) */
  void* (*init)(void* user, int thread, int n_threads);
  /* ONE pass over this thread's share of the items */
  void (*work)(void* user, void* state, int thread, int n_threads);
  void (*fini)(void* user, void* state);
  /* optional, called by thread 0 between two passes' barriers (e.g. to reset a shared work counter); with it set, every
   * pass is bracketed by barriers (dynamic scheduling); without it the R passes of a thread run back to back */
  void (*between)(void* user);
} pxo_bench_ops;

typedef struct {
  double seconds;        /* timed region (thread 0's clock between the barrier releases) */
  int64_t passes;        /* passes inside it */
  double calib_seconds;  /* the untimed calibration pass */
  int n_threads;
  int pinned;            /* threads were pinned to CPUs */
} pxo_bench_result;

typedef struct pxo_bench_shared_ {
  const pxo_bench_ops* ops; void* user;
  int n_threads; double min_seconds; int64_t max_passes;
  pthread_barrier_t bar;
  volatile int64_t passes;           /* decided by thread 0 after the calibration pass */
  double t_calib, t_timed;
  int n_cpus; int cpus[4096]; int pin;
} pxo_bench_shared;

typedef struct { pxo_bench_shared* sh; int thread; } pxo_bench_arg;

static double pxo_bench_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* pxo_bench_thread(void* argp) {
  pxo_bench_arg* arg = (pxo_bench_arg*)argp;
  pxo_bench_shared* sh = arg->sh;
  const int t = arg->thread, T = sh->n_threads;
  if (sh->pin && sh->n_cpus > 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(sh->cpus[t % sh->n_cpus], &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
  }
  void* state = sh->ops->init ? sh->ops->init(sh->user, t, T) : NULL;
  /* calibration pass (also the warm-up: page faults, caches, clock ramp) */
  pthread_barrier_wait(&sh->bar);
  double t0 = t == 0 ? pxo_bench_now() : 0.0;
  sh->ops->work(sh->user, state, t, T);
  pthread_barrier_wait(&sh->bar);
  if (t == 0) {
    sh->t_calib = pxo_bench_now() - t0;
    int64_t r = 1;
    if (sh->min_seconds > 0.0 && sh->t_calib > 0.0) r = (int64_t)(sh->min_seconds / sh->t_calib) + 1;
    if (r > sh->max_passes) r = sh->max_passes;
    if (r < 1) r = 1;
    sh->passes = r;
    if (sh->ops->between) sh->ops->between(sh->user);
  }
  pthread_barrier_wait(&sh->bar);
  const int64_t R = sh->passes;
  if (t == 0) t0 = pxo_bench_now();
  if (sh->ops->between) {
    for (int64_t r = 0; r < R; ++r) {
      sh->ops->work(sh->user, state, t, T);
      pthread_barrier_wait(&sh->bar);
      if (t == 0 && r + 1 < R) sh->ops->between(sh->user);
      if (r + 1 < R) pthread_barrier_wait(&sh->bar);
    }
  } else {
    for (int64_t r = 0; r < R; ++r) sh->ops->work(sh->user, state, t, T);
    pthread_barrier_wait(&sh->bar);
  }
  if (t == 0) sh->t_timed = pxo_bench_now() - t0;
  if (sh->ops->fini) sh->ops->fini(sh->user, state);
  return NULL;
}

/* 0 on success.  min_seconds <= 0: one timed pass (after the calibration / warm-up pass). */
static int pxo_bench_run(const pxo_bench_ops* ops, void* user, int n_threads, double min_seconds, int64_t max_passes,
                         pxo_bench_result* out) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 4096) n_threads = 4096;
  pxo_bench_shared* sh = (pxo_bench_shared*)calloc(1, sizeof(pxo_bench_shared));
  if (!sh) return -1;
  sh->ops = ops; sh->user = user; sh->n_threads = n_threads; sh->min_seconds = min_seconds;
  sh->max_passes = max_passes > 0 ? max_passes : 1000000;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  sh->pin = getenv("PXO_BENCH_NO_PIN") == NULL && sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
  if (sh->pin)
    for (int c = 0; c < CPU_SETSIZE && sh->n_cpus < 4096; ++c)
      if (CPU_ISSET(c, &allowed)) sh->cpus[sh->n_cpus++] = c;
  if (pthread_barrier_init(&sh->bar, NULL, (unsigned)n_threads) != 0) { free(sh); return -1; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
  pxo_bench_arg* args = (pxo_bench_arg*)malloc(sizeof(pxo_bench_arg) * (size_t)n_threads);
  int started = 0;
  for (int t = 0; t < n_threads; ++t) {
    args[t].sh = sh; args[t].thread = t;
    if (pthread_create(&th[t], NULL, pxo_bench_thread, &args[t]) != 0) break;
    ++started;
  }
  if (started != n_threads) {      /* cannot release the barrier: the started threads are parked for good; give up loudly */
    for (int t = 0; t < started; ++t) pthread_cancel(th[t]);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&sh->bar); free(th); free(args); free(sh);
    return -2;
  }
  for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  out->seconds = sh->t_timed; out->passes = sh->passes; out->calib_seconds = sh->t_calib;
  out->n_threads = n_threads; out->pinned = sh->pin && sh->n_cpus > 0;
  pthread_barrier_destroy(&sh->bar);
  free(th); free(args); free(sh);
  return 0;
}

/* contiguous share of thread t of n items */
static void pxo_bench_share(int64_t n, int t, int T, int64_t* first, int64_t* count) {
  const int64_t lo = n * t / T, hi = n * (t + 1) / T;
  *first = lo; *count = hi - lo;
}
#endif /* PXO_BENCH_HARNESS_H_ */
