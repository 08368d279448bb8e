// Replay of a recorded sequence of C-ABI calls in ONE call from the interpreter (round 5; _hip.planned): the teacher's no-grad backbone
// pass is ~150 launches whose arguments do not change from step to step.  Replaying them from a Python loop costs the teacher's host
// thread 3 ms per step in which it holds -- and hands back and forth -- the interpreter lock the step thread needs for the student's
// heads; from here the whole sequence is issued with the lock released.  Every recorded entry point takes integer / pointer
// arguments only (at most 16) and returns int: it is called through one 16-argument prototype (System V x86-64: surplus
// integer arguments are ignored by the callee).
#include "common.h"

typedef int (*mmt_fn16)(long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long);

extern "C" int mmt_replay(const mmt_call* calls, int n, int* failed_index) {
  if (!calls || n < 0) return MMT_EINVAL;
  for (int i = 0; i < n; i++) {
    const long* a = calls[i].a;
    const int rc = ((mmt_fn16)calls[i].fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
    if (rc) {
      if (failed_index) *failed_index = i;
      return rc;
    }
  }
  return 0;
}
