// tip_rnnh.h — recurrence with the output projection inside its hop wait (tip_rnnh.hip)
#pragma once
#include "tip_internal.h"

namespace tip {

// paper recurrence (rnn_hidden 512) with 128 < size_s <= 132 outputs; any window length the ring's step tags cover
bool rnnh_supported(const Dims& d, int T);
// bytes of the hand-off ring (two 16-KB slots per four-workgroup cluster in flight) for a batch of B windows
size_t rnnh_ring_bytes(int B);
// y [B][T][S] (or [B][S], row T-1, when last_only) = (tanh recurrence over ih [B][T][512]) W_out^T + b.  wout: W_out row-major
// [>= 132 rows, zero padded][512]; ring: rnnh_ring_bytes() of workspace, filled with 0xFF by whoever ran in front when
// ring_armed, else by a memset here; flags: the XCC-exchange words the other recurrence kernels use; etag: the launch tag.
hipError_t launch_rnn_head(const Dims& d, const float* ih, const float* whh_frag, const float* wout, const float* bout, float* y,
                           float* ring, unsigned* flags, int B, int T, bool last_only, bool ring_armed, int num_cus, unsigned etag,
                           const Guard& gd, hipStream_t s);
hipError_t read_spin_timeouts_rnnh(unsigned* out);

}  // namespace tip
