/*
 * jvector-native-hip: VectorizationProvider backed by libjvector_hip.so (MI355X / gfx950).
 * Template: jvector-native/.../vector/NativeVectorizationProvider.java:28-53 of the reference.
 * NOT compiled in this repository (no JDK in the build image) — see jvector-native-hip/README.md.
 */
package io.github.jbellis.jvector.vector;

import io.github.jbellis.jvector.vector.hip.HipOps;
import io.github.jbellis.jvector.vector.types.VectorTypeSupport;

public class HipVectorizationProvider extends VectorizationProvider {
    private final VectorUtilSupport vectorUtilSupport;
    private final VectorTypeSupport vectorTypeSupport;

    public HipVectorizationProvider() {
        // Same fall-back signalling as the reference: throw UnsupportedOperationException and
        // VectorizationProvider.lookup() degrades to Panama/Default (VectorizationProvider.java:141-149).
        if (!HipOps.load()) {
            throw new UnsupportedOperationException("Failed to load libjvector_hip");
        }
        if (HipOps.deviceCount() <= 0) {
            throw new UnsupportedOperationException("No gfx950 device visible: " + HipOps.lastError());
        }
        this.vectorUtilSupport = new HipVectorUtilSupport();
        // identical storage types to the native provider: MemorySegment-backed vectors / byte sequences
        this.vectorTypeSupport = new MemorySegmentVectorProvider();
    }

    @Override
    public VectorUtilSupport getVectorUtilSupport() {
        return vectorUtilSupport;
    }

    @Override
    public VectorTypeSupport getVectorTypeSupport() {
        return vectorTypeSupport;
    }
}
