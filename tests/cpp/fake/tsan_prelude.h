// ThreadSanitizer builds only (tests/test_sanitizers.py, -include): GCC 11's libtsan does not intercept pthread_cond_clockwait, which libstdc++
// uses for condition_variable::wait_for on the steady clock -- TSan then misses the unlock / lock inside the wait and reports every access
// under that mutex as a race.  Without the macro libstdc++ falls back to pthread_cond_timedwait, which IS intercepted.
#include <bits/c++config.h>
#undef _GLIBCXX_USE_PTHREAD_COND_CLOCKWAIT
