/* mmo_batch.c -- multi-env / multi-thread driver around the fp64 oracle, used by
 * bench.py's cpu_baseline leg and by parity tests that need many envs.
 * TEST INFRASTRUCTURE ONLY (see mmo_engine.c header).
 *
 * One env-step here = what the reference's env.step does at the engine level
 * (myosuite/envs/myo/base_v0.py:82-118 -> robot/robot.py:856-861,595-607):
 * optional muscle ctrl map sigma(5(a-0.5)), `nsub` mj_step substeps, and one
 * mj_forward on the new state.                                              */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct mmo_model mmo_model;
typedef struct mmo_data mmo_data;
mmo_data* mmo_data_create(const mmo_model* m);
void mmo_data_free(mmo_data* d);
void mmo_step(const mmo_model* m, mmo_data* d);
void mmo_forward(const mmo_model* m, mmo_data* d);
double* mmo_field(mmo_data* d, const char* name);
int mmo_dim(const mmo_model* m, int which);

typedef struct {
  const mmo_model* m; mmo_data** d; int e0, e1, nsub, nsteps, nu, normalize, do_forward;
  const double* actions; /* [nsteps][nenv][nu] */ int nenv;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  for (int s = 0; s < j->nsteps; s++)
    for (int e = j->e0; e < j->e1; e++) {
      double* ctrl = mmo_field(j->d[e], "ctrl");
      const double* a = j->actions + ((size_t)s * j->nenv + e) * j->nu;
      for (int k = 0; k < j->nu; k++) ctrl[k] = j->normalize ? 1.0 / (1.0 + exp(-5.0 * (a[k] - 0.5))) : a[k];
      for (int k = 0; k < j->nsub; k++) mmo_step(j->m, j->d[e]);
      if (j->do_forward) mmo_forward(j->m, j->d[e]);
    }
  return NULL;
}

/* advance `nenv` independent envs by `nsteps` env-steps on `nthreads` threads */
void mmo_batch_rollout(const mmo_model* m, mmo_data** d, int nenv, int nthreads, int nsub, int nsteps,
                       const double* actions, int normalize, int do_forward) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > nenv) nthreads = nenv;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * nthreads);
  int nu = mmo_dim(m, 2);
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (job_t){m, d, (int)((long)nenv * t / nthreads), (int)((long)nenv * (t + 1) / nthreads), nsub,
                      nsteps, nu, normalize, do_forward, actions, nenv};
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
}
