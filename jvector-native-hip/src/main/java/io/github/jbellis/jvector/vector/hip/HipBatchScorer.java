/*
 * Batched call sites: what a lock-step multi-query searcher / builder calls instead of the per-node SPI.  Every row of
 * INTEGRATION.md's call-site table has a method here:
 *
 *   reference call site (per node / per pair)                                         batched method
 *   PQDecoder / FusedPQDecoder constructors        (PQDecoder.java:41-54,88-122)      prepare
 *   ScoreFunction.similarityTo(node)               (ScoreFunction.java:41)            similarityTo
 *   ScoreFunction.similarityToNeighbor(origin, i)  (FusedPQDecoder.java:104-111)      similarityToNeighbors
 *   NodeQueue.rerank -> ExactScoreFunction         (NodeQueue.java:160-195)           rerank
 *   NodeQueue / BoundedLongHeap ordering           (NodeQueue.java:125-129)           topK
 *   GraphSearcher.search                           (GraphSearcher.java:222-243)       search
 *   ProductQuantization.encodeAll / PQVectors.encodeAndBuild (PQVectors.java:109-152) encodeAll
 *   FusedPQ.writeInline                            (FusedPQ.java:146-161)             attachFusedFromCodes
 *   ImmutablePQVectors.diversityFunctionFor        (ImmutablePQVectors.java:61-104)   diversityScores
 *
 * One instance per searcher thread (a jv_ctx is single-threaded).  All buffers are OFF-HEAP MemorySegments owned by the caller.
 * NOT compiled in this repository (no JDK in the build image).
 */
package io.github.jbellis.jvector.vector.hip;

import io.github.jbellis.jvector.vector.VectorSimilarityFunction;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import static java.lang.foreign.ValueLayout.JAVA_FLOAT;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

public final class HipBatchScorer implements AutoCloseable {
    /** jv_decoder_kind: which reference decoder's cosine query-magnitude arithmetic to reproduce */
    public static final int DECODER_PQ = 0, DECODER_FUSED = 1;
    public static final int TRAVERSAL_AUTO = 0, TRAVERSAL_HOST = 1, TRAVERSAL_DEVICE = 2;

    private final Arena arena = Arena.ofConfined();
    private final MemorySegment ctx;
    private MemorySegment pq, codes, luts, vectors = MemorySegment.NULL, fused = MemorySegment.NULL, graph = MemorySegment.NULL,
            pairTable = MemorySegment.NULL;
    private int maxQueries;
    private VectorSimilarityFunction vsf;

    public HipBatchScorer(int device) {
        this.ctx = HipOps.ctxCreate(arena, device);
    }

    /** pqBytes: ProductQuantization.write output (B/quantization/ProductQuantization.java:560-599), off-heap. */
    public void attach(MemorySegment pqBytes, MemorySegment codeBytes, long count, int maxQueries) {
        this.pq = HipOps.pqLoad(arena, ctx, pqBytes);
        this.codes = HipOps.codesCreate(arena, ctx, pq, count);
        if (codeBytes != null) HipOps.codesUpload(ctx, codes, 0, count, codeBytes);
        this.luts = HipOps.lutsCreate(arena, ctx, pq, maxQueries);
        this.maxQueries = maxQueries;
    }

    /** RandomAccessVectorValues for the reranker: count x dim floats, row-major, little-endian, off-heap. */
    public void attachVectors(MemorySegment rows, long count, int dim) {
        this.vectors = HipOps.vectorsCreate(arena, ctx, count, dim);
        HipOps.vectorsUpload(ctx, vectors, 0, count, rows);
    }

    /** FusedPQ inline blocks as stored in the index (FusedPQ.writeInline layout) + the matching level-0 adjacency. */
    public void attachFused(MemorySegment blocks, MemorySegment neighbors, long count, int maxDegree) {
        this.fused = HipOps.fusedCreate(arena, ctx, pq, count, maxDegree);
        HipOps.fusedUpload(ctx, fused, 0, count, blocks, neighbors);
    }

    /** FusedPQ.writeInline on the device: the blocks are gathered from the attached code store. */
    public void attachFusedFromCodes(MemorySegment neighbors, long count, int maxDegree) {
        this.fused = HipOps.fusedCreate(arena, ctx, pq, count, maxDegree);
        HipOps.fusedBuild(ctx, fused, codes, 0, count, neighbors);
    }

    /** level 0 holds every node (nodeIds == null); upper levels list ascending node ids; rows are -1 padded. */
    public void attachGraph(long nNodes, int nLevels, int entryNode, int entryLevel, MemorySegment[] nodeIds, MemorySegment[] neighbors,
                            int[] counts, int[] degrees) {
        this.graph = HipOps.graphCreate(arena, ctx, nNodes, nLevels);
        for (int l = 0; l < nLevels; l++)
            HipOps.graphSetLevel(ctx, graph, l, counts[l], nodeIds[l] == null ? MemorySegment.NULL : nodeIds[l], neighbors[l], degrees[l]);
        HipOps.graphSetEntry(graph, entryNode, entryLevel);
    }

    public void setTraversal(int mode) { HipOps.graphSetTraversal(graph, mode); }

    /** PQDecoder / FusedPQDecoder constructors for Q queries at once. queries: Q x D floats off-heap. */
    public void prepare(MemorySegment queries, int q, VectorSimilarityFunction vsf, int decoderKind) {
        if (q > maxQueries) throw new IllegalArgumentException("batch of " + q + " queries exceeds the capacity " + maxQueries);
        this.vsf = vsf;
        HipOps.lutsBuild(ctx, luts, queries, q, vsf.ordinal(), decoderKind);
    }

    public void prepare(MemorySegment queries, int q, VectorSimilarityFunction vsf) { prepare(queries, q, vsf, DECODER_PQ); }

    /** scores[q*b + j] = similarityTo(ordinals[q*b + j]); ordinal -1 = empty slot (-Infinity). */
    public void similarityTo(MemorySegment ordinals, int b, MemorySegment scoresOut) {
        HipOps.adcScores(ctx, luts, codes, ordinals, b, scoresOut);
    }

    /** scoresOut[q*maxDegree + i] = similarityToNeighbor(origins[q], i) for every prepared query; neighborsOut nullable. */
    public void similarityToNeighbors(MemorySegment origins, MemorySegment scoresOut, MemorySegment neighborsOutOrNull) {
        HipOps.fusedScores(ctx, luts, fused, origins, scoresOut, neighborsOutOrNull == null ? MemorySegment.NULL : neighborsOutOrNull);
    }

    /** exact scores of candidates[q*b + j] against queries[q] (the reranker's ExactScoreFunction) */
    public void rerank(MemorySegment queries, int q, VectorSimilarityFunction vsf, MemorySegment candidates, int b, MemorySegment scoresOut) {
        HipOps.exactScores(ctx, vectors, queries, q, vsf.ordinal(), candidates, b, scoresOut);
    }

    /** NodeQueue order: higher score first, ties -> smaller node id; (-1, -Infinity) padded */
    public void topK(MemorySegment scores, MemorySegment idsOrNull, int q, long n, int k, MemorySegment outIds, MemorySegment outScores) {
        HipOps.topk(ctx, scores, idsOrNull == null ? MemorySegment.NULL : idsOrNull, q, n, n, 0, k, outIds, outScores);
    }

    /** GraphSearcher.search for the whole batch: ids / scores are q x topK; stats (nullable) q x {visited, expanded} int64;
     *  acceptBits (nullable): Bits as a little-endian bit array over node ids, stride 0 = one mask for the batch. */
    public void search(MemorySegment queries, int q, VectorSimilarityFunction vsf, int topK, int rerankK, MemorySegment acceptBitsOrNull,
                       long acceptStrideWords, MemorySegment outIds, MemorySegment outScores, MemorySegment statsOrNull) {
        HipOps.graphSearchFiltered(ctx, graph, luts, codes, fused, vectors, queries, q, vsf.ordinal(), topK, rerankK,
                                   acceptBitsOrNull == null ? MemorySegment.NULL : acceptBitsOrNull, acceptStrideWords, outIds, outScores,
                                   statsOrNull == null ? MemorySegment.NULL : statsOrNull);
    }

    /** two-pass search without a graph (ADC scan of every code -> top rerankK -> exact rerank -> topK) */
    public void searchFlat(MemorySegment queries, int q, VectorSimilarityFunction vsf, int topK, int rerankK, MemorySegment outIds, MemorySegment outScores) {
        HipOps.searchFlat(ctx, luts, codes, vectors, queries, q, vsf.ordinal(), topK, rerankK, 0, outIds, outScores);
    }

    /** ProductQuantization.encodeAll: count x D floats -> count x M code bytes (first minimum wins, NaN never wins) */
    public void encodeAll(MemorySegment rows, long count, MemorySegment codesOut) {
        HipOps.pqEncode(ctx, pq, rows, count, codesOut);
    }

    /** BuildScoreProvider's diversity provider: out[p*b + j] = similarity(code[node1[p]], code[node2[p*b + j]]) */
    public void diversityScores(VectorSimilarityFunction vsf, MemorySegment node1, int p, MemorySegment node2, int b, MemorySegment out) {
        if (pairTable.equals(MemorySegment.NULL)) pairTable = HipOps.pairTableCreate(arena, ctx, pq, vsf.ordinal());
        HipOps.codePairScores(ctx, pairTable, codes, node1, p, node2, b, out);
    }

    public void sync() { HipOps.ctxSync(ctx); }

    public static MemorySegment allocFloats(Arena a, long n) { return a.allocate(JAVA_FLOAT, n); }
    public static MemorySegment allocInts(Arena a, long n) { return a.allocate(JAVA_INT, n); }
    public static MemorySegment allocLongs(Arena a, long n) { return a.allocate(JAVA_LONG, n); }

    @Override
    public void close() {
        if (!pairTable.equals(MemorySegment.NULL)) HipOps.pairTableDestroy(pairTable);
        if (!graph.equals(MemorySegment.NULL)) HipOps.graphDestroy(graph);
        if (!fused.equals(MemorySegment.NULL)) HipOps.fusedDestroy(fused);
        if (!vectors.equals(MemorySegment.NULL)) HipOps.vectorsDestroy(vectors);
        if (luts != null) HipOps.lutsDestroy(luts);
        if (codes != null) HipOps.codesDestroy(codes);
        if (pq != null) HipOps.pqDestroy(pq);
        HipOps.ctxDestroy(ctx);
        arena.close();
    }
}
