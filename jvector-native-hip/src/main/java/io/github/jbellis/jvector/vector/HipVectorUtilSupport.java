/*
 * Per-pair SPI of the HIP provider.  libjvector_hip.so exports the reference's 24 native symbols unchanged
 * (include/jvector_simd_compat.h), so the reference's own binding class works as is: this subclass only exists to
 * (a) name the provider and (b) hang the batched entry points off the same object for call sites that opt in.
 * Template: jvector-native/.../vector/NativeVectorUtilSupport.java:36-299.
 * NOT compiled in this repository (no JDK in the build image).
 */
package io.github.jbellis.jvector.vector;

import io.github.jbellis.jvector.vector.hip.HipBatchScorer;
import io.github.jbellis.jvector.vector.hip.HipOps;

final class HipVectorUtilSupport extends NativeVectorUtilSupport {
    HipVectorUtilSupport() {
        super(); // NativeSimdOps' SymbolLookup.loaderLookup() resolves cosine_f32 ... in libjvector_hip.so
    }

    /** "gfx950:sramecc+:xnack-" — the analogue of NativeVectorUtilSupport.getActiveIsa() (:45-49) */
    public String getActiveArch() {
        return HipOps.activeArch(0);
    }

    /** Batched scorer bound to one device context; one per searcher thread (contexts are not thread-safe). */
    public HipBatchScorer newBatchScorer(int device) {
        return new HipBatchScorer(device);
    }
}
