/* A caller that OWNS the Runge-Kutta loop, in C: the execution shape of
 * integrate.odeint (integrate.py:143-169 -- the driver calls the right-hand side
 * once per stage) with the explicit midpoint rule of model.integrate_ode
 * (model.py:155-157), written against include/ddd1d.h only.  Two fused launches
 * per step through ddd_rk_substep; with `chained` the loop is bracketed by
 * ddd_stream_fork / ddd_stream_join.  bench.py times this loop
 * (configs.rk_substep_external) next to the same loop written in Python, and
 * tests/test_gpu_integrate.py checks it bit for bit against ddd_integrate_fixed.
 * Built by __graft_entry__.build() with gcc into examples/librk_driver.so. */
#include <stddef.h>

#include "../include/ddd1d.h"

#if defined(__GNUC__)
__attribute__((visibility("default")))
#endif
int rk_driver_midpoint(ddd_model* model, int steps, double t0, double dt, float* y, float* ystage,
                       float* ynew, int batch, void* stream, int chained, float** final_state) {
  const float h = (float)dt;
  int rc = chained ? ddd_stream_fork(model, stream) : 0;
  for (int step = 0; step < steps && rc == 0; ++step) {
    const double t = t0 + (double)step * dt;
    /* stage 1: ystage = y + h/2 f(t, y);  stage 2: ynew = y + h f(t + h/2, ystage) */
    rc = ddd_rk_substep(model, t, y, y, 0.5f * h, ystage, NULL, 0.0f, NULL, batch, stream);
    if (rc == 0)
      rc = ddd_rk_substep(model, t + 0.5 * dt, ystage, NULL, 0.0f, NULL, y, h, ynew, batch, stream);
    float* swap = y; y = ynew; ynew = swap;
  }
  if (chained) {
    const int rc_join = ddd_stream_join(model, stream);
    if (rc == 0) rc = rc_join;
  }
  if (final_state != NULL) *final_state = y;
  return rc;
}
