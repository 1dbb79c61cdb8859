// Host-side step sequencer: a recorded list of stream operations (captured-graph launches, event records / waits, batched
// staging copies) executed by ONE call.
//
// The reference's rollout step is a serial chain with two host round trips in it (ss_baselines/savi/ppo/ppo_trainer.py:449-636:
// act_option -> the host reads the option actions -> Speaker / clip.tokenize -> act / act_dialog -> envs.step).  On this path every
// forward is a captured HIP graph whose launch sequence is fixed once the argument buffers are known, so what the host does
// between "the option actions have landed" and "the text tower's first kernel is in the queue" should be one function call, not
// a page of Python (events created per call, stream context managers, ctypes marshalling per launch): the Python layer
// (avlen_amd/sequencer.py) builds the list once per set of argument buffers and replays it here.
#include "common.h"
#include "../../include/avlen_hip.h"

extern "C" int avlen_cmds_run(const avlen_cmd* cmds, int n) {
  if (n < 0 || (n > 0 && !cmds)) return AVLEN_ERR_ARG;
  for (int i = 0; i < n; i++) {
    const avlen_cmd& c = cmds[i];
    hipError_t e = hipSuccess;
    switch (c.op) {
      case AVLEN_CMD_GRAPH:
        if (!c.a) return AVLEN_ERR_ARG;
        e = hipGraphLaunch((hipGraphExec_t)c.a, (hipStream_t)c.b);
        break;
      case AVLEN_CMD_RECORD:
        if (!c.a) return AVLEN_ERR_ARG;
        e = hipEventRecord((hipEvent_t)c.a, (hipStream_t)c.b);
        break;
      case AVLEN_CMD_WAIT:
        if (!c.b) return AVLEN_ERR_ARG;
        e = hipStreamWaitEvent((hipStream_t)c.a, (hipEvent_t)c.b, 0);
        break;
      case AVLEN_CMD_MULTICOPY: {
        const int rc = avlen_multi_copy((const void* const*)c.a, (void* const*)c.b, (const int64_t*)c.c, c.n, (hipStream_t)c.d);
        if (rc != AVLEN_OK) return rc;
        break;
      }
      default:
        return AVLEN_ERR_ARG;
    }
    if (e != hipSuccess) return AVLEN_ERR_LAUNCH;
  }
  return AVLEN_OK;
}
