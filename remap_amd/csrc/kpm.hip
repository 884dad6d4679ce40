// kpm.hip — placeholder, filled in next
#include "common.h"
namespace rmx {
struct KpmWorkspace { void* p = nullptr; size_t bytes = 0; };
inline size_t kpm_workspace_bytes(const KpmWorkspace& w) { return w.bytes; }
inline void kpm_workspace_free(KpmWorkspace& w) { if (w.p) (void)hipFree(w.p); w.p = nullptr; w.bytes = 0; }
}
