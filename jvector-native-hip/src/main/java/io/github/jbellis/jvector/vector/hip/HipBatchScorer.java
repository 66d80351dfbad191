/*
 * Batched call sites: what a lock-step multi-query searcher calls instead of per-node
 * ScoreFunction.similarityTo (B/graph/similarity/ScoreFunction.java:41,52).  One instance per searcher thread.
 * NOT compiled in this repository (no JDK in the build image).
 */
package io.github.jbellis.jvector.vector.hip;

import io.github.jbellis.jvector.vector.VectorSimilarityFunction;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import static java.lang.foreign.ValueLayout.JAVA_FLOAT;
import static java.lang.foreign.ValueLayout.JAVA_INT;

public final class HipBatchScorer implements AutoCloseable {
    private final Arena arena = Arena.ofConfined();
    private final MemorySegment ctx;
    private MemorySegment pq, codes, luts;
    private int maxQueries;

    public HipBatchScorer(int device) {
        this.ctx = HipOps.ctxCreate(arena, device);
    }

    /** pqBytes: ProductQuantization.write output (B/quantization/ProductQuantization.java:560-599), off-heap. */
    public void attach(MemorySegment pqBytes, MemorySegment codeBytes, long count, int maxQueries) {
        this.pq = HipOps.pqLoad(arena, ctx, pqBytes);
        this.codes = HipOps.codesCreate(arena, ctx, pq, count);
        HipOps.codesUpload(ctx, codes, 0, count, codeBytes);
        this.luts = HipOps.lutsCreate(arena, ctx, pq, maxQueries);
        this.maxQueries = maxQueries;
    }

    /** PQDecoder constructors for Q queries at once (PQDecoder.java:41-54,88-122). queries: Q x D floats off-heap. */
    public void prepare(MemorySegment queries, int q, VectorSimilarityFunction vsf) {
        HipOps.lutsBuild(ctx, luts, queries, q, vsf.ordinal(), 0);
    }

    /** scores[q*b + j] = similarityTo(ordinals[q*b + j]); ordinal -1 = empty slot (-Infinity). */
    public void similarityTo(MemorySegment ordinals, int b, MemorySegment scoresOut) {
        HipOps.adcScores(ctx, luts, codes, ordinals, b, scoresOut);
    }

    public static MemorySegment allocFloats(Arena a, long n) { return a.allocate(JAVA_FLOAT, n); }
    public static MemorySegment allocInts(Arena a, long n) { return a.allocate(JAVA_INT, n); }

    @Override
    public void close() {
        HipOps.ctxDestroy(ctx);
        arena.close();
    }
}
