/*
 * GoldenDump — run with a REAL JVector (this repository's reference) to pin the MI355X engine's oracle at the literal-value level.
 *
 * The build image of the HIP engine has no JDK, so its CPU oracle (oracle/jv_oracle.c) is a line-cited restatement of the
 * reference, pinned only where the reference ships literal fixtures.  This program produces the missing literals with the
 * reference's own classes under the scalar provider (-Djvector.vectorization_provider=default ... or simply no Panama/native
 * modules on the class path): PQ code bytes, ADC / direct / diversity scores as raw float bits, robust-prune selections, a built
 * graph with search results and counters, an on-disk index (v6, FusedPQ + inline vectors) plus a PQVectors blob, and NVQ (global
 * mean, encoded bytes + the four parameters per sub-vector, NVQScorer scores, the NVQVectors blob) for 1 and 3 sub-vectors.
 * tests/test_reference_goldens.py consumes the file (tests/golden/ref/jvector_goldens.bin) and checks BOTH the oracle and the
 * HIP library against it.
 *
 *   mvn -q -pl jvector-native-hip -am test-compile
 *   mvn -q -pl jvector-native-hip exec:java -Dexec.classpathScope=test \
 *       -Dexec.mainClass=io.github.jbellis.jvector.vector.hip.GoldenDump -Dexec.args="/path/to/repo/tests/golden/ref/jvector_goldens.bin"
 *
 * Container (little-endian): magic "JVGOLD01", then records { u32 nameLen, name utf-8, u8 dtype (0 u8, 1 i32, 2 f32, 3 i64),
 * u32 ndim, u32 dims[ndim], payload }.  NOT compiled in the engine's own build environment (no JDK there).
 */
package io.github.jbellis.jvector.vector.hip;

import io.github.jbellis.jvector.disk.SimpleMappedReader;
import io.github.jbellis.jvector.disk.SimpleWriter;
import io.github.jbellis.jvector.graph.GraphIndexBuilder;
import io.github.jbellis.jvector.graph.GraphSearcher;
import io.github.jbellis.jvector.graph.ImmutableGraphIndex;
import io.github.jbellis.jvector.graph.ListRandomAccessVectorValues;
import io.github.jbellis.jvector.graph.NodeArray;
import io.github.jbellis.jvector.graph.SearchResult;
import io.github.jbellis.jvector.graph.disk.OnDiskGraphIndex;
import io.github.jbellis.jvector.graph.disk.OnDiskGraphIndexWriter;
import io.github.jbellis.jvector.graph.disk.feature.Feature;
import io.github.jbellis.jvector.graph.disk.feature.FeatureId;
import io.github.jbellis.jvector.graph.disk.feature.FusedPQ;
import io.github.jbellis.jvector.graph.disk.feature.InlineVectors;
import io.github.jbellis.jvector.graph.diversity.VamanaDiversityProvider;
import io.github.jbellis.jvector.graph.similarity.BuildScoreProvider;
import io.github.jbellis.jvector.graph.similarity.DefaultSearchScoreProvider;
import io.github.jbellis.jvector.quantization.NVQVectors;
import io.github.jbellis.jvector.quantization.NVQuantization;
import io.github.jbellis.jvector.quantization.PQVectors;
import io.github.jbellis.jvector.quantization.ProductQuantization;
import io.github.jbellis.jvector.util.BitSet;
import io.github.jbellis.jvector.util.Bits;
import io.github.jbellis.jvector.util.FixedBitSet;
import io.github.jbellis.jvector.vector.VectorSimilarityFunction;
import io.github.jbellis.jvector.vector.VectorizationProvider;
import io.github.jbellis.jvector.vector.types.VectorFloat;
import io.github.jbellis.jvector.vector.types.VectorTypeSupport;

import java.io.ByteArrayOutputStream;
import java.io.DataOutputStream;
import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Path;
import java.util.ArrayList;
import java.util.EnumMap;
import java.util.List;
import java.util.function.IntFunction;

public final class GoldenDump {
    private static final VectorTypeSupport VTS = VectorizationProvider.getInstance().getVectorTypeSupport();

    // ---- seeded inputs: splitmix64 -> uniform in [-1, 1), the generator tests/test_reference_goldens.py restates ----
    private static long state;
    private static long next() {
        long z = (state += 0x9E3779B97F4A7C15L);
        z = (z ^ (z >>> 30)) * 0xBF58476D1CE4E5B9L;
        z = (z ^ (z >>> 27)) * 0x94D049BB133111EBL;
        return z ^ (z >>> 31);
    }
    private static float uniform() { return (float) ((next() >>> 40) / (double) (1L << 24)) * 2.0f - 1.0f; }

    private static float[][] matrix(long seed, int n, int d, boolean clustered) {
        state = seed;
        float[][] centers = new float[16][d];
        for (float[] c : centers) for (int j = 0; j < d; j++) c[j] = uniform();
        float[][] m = new float[n][d];
        for (int i = 0; i < n; i++) {
            float[] c = centers[(int) ((next() >>> 33) % 16)];
            for (int j = 0; j < d; j++) m[i][j] = (clustered ? c[j] : 0.0f) + 0.35f * uniform();
        }
        return m;
    }

    // ---- container ----
    private final DataOutputStream out;
    private GoldenDump(DataOutputStream out) { this.out = out; }
    private void header(String name, int dtype, int... dims) throws IOException {
        byte[] nb = name.getBytes(StandardCharsets.UTF_8);
        ByteBuffer b = ByteBuffer.allocate(4 + nb.length + 1 + 4 + 4 * dims.length).order(ByteOrder.LITTLE_ENDIAN);
        b.putInt(nb.length).put(nb).put((byte) dtype).putInt(dims.length);
        for (int d : dims) b.putInt(d);
        out.write(b.array());
    }
    private void bytes(String name, byte[] v, int... dims) throws IOException { header(name, 0, dims.length == 0 ? new int[]{v.length} : dims); out.write(v); }
    private void ints(String name, int[] v, int... dims) throws IOException {
        header(name, 1, dims.length == 0 ? new int[]{v.length} : dims);
        ByteBuffer b = ByteBuffer.allocate(4 * v.length).order(ByteOrder.LITTLE_ENDIAN);
        for (int x : v) b.putInt(x);
        out.write(b.array());
    }
    private void floats(String name, float[] v, int... dims) throws IOException {
        header(name, 2, dims.length == 0 ? new int[]{v.length} : dims);
        ByteBuffer b = ByteBuffer.allocate(4 * v.length).order(ByteOrder.LITTLE_ENDIAN);
        for (float x : v) b.putInt(Float.floatToRawIntBits(x));   // raw bits: NaN payloads survive
        out.write(b.array());
    }

    private static float[] flat(float[][] m) {
        float[] f = new float[m.length * m[0].length];
        for (int i = 0; i < m.length; i++) System.arraycopy(m[i], 0, f, i * m[0].length, m[0].length);
        return f;
    }

    private static List<VectorFloat<?>> vectors(float[][] m) {
        List<VectorFloat<?>> l = new ArrayList<>(m.length);
        for (float[] r : m) l.add(VTS.createFloatVector(r.clone()));
        return l;
    }

    public static void main(String[] args) throws Exception {
        Path target = Path.of(args.length > 0 ? args[0] : "jvector_goldens.bin");
        final int N = 2000, D = 64, M = 8, Q = 16, DEG = 16, BEAM = 40, TOPK = 10, RERANK = 40;
        float[][] base = matrix(1234, N, D, true), queries = matrix(99, Q, D, true);
        var ravv = new ListRandomAccessVectorValues(vectors(base), D);
        List<VectorFloat<?>> qv = vectors(queries);

        Files.createDirectories(target.toAbsolutePath().getParent());
        try (var dos = new DataOutputStream(Files.newOutputStream(target))) {
            dos.write("JVGOLD01".getBytes(StandardCharsets.US_ASCII));
            GoldenDump g = new GoldenDump(dos);
            g.ints("shape", new int[]{N, D, M, Q, DEG, BEAM, TOPK, RERANK});
            g.bytes("provider", VectorizationProvider.getInstance().getClass().getSimpleName().getBytes(StandardCharsets.UTF_8));
            g.floats("vectors", flat(base), N, D);
            g.floats("queries", flat(queries), Q, D);

            // ---- ProductQuantization: trained here (nondeterministic), shipped as its own wire bytes; everything below is a pure
            //      function of those bytes and the inputs above ----
            ProductQuantization pq = ProductQuantization.compute(ravv, M, 256, false);
            Path tmp = Files.createTempFile("golden", ".pq");
            try (var w = new SimpleWriter(tmp)) { pq.write(w, OnDiskGraphIndex.CURRENT_VERSION); }
            g.bytes("pq_bytes", Files.readAllBytes(tmp));
            PQVectors pqv = (PQVectors) pq.encodeAll(ravv);
            byte[] codes = new byte[N * M];
            for (int i = 0; i < N; i++) {
                var c = pqv.get(i);
                for (int m = 0; m < M; m++) codes[i * M + m] = c.get(m);
            }
            g.bytes("codes", codes, N, M);                                                       // ProductQuantization.encode, row 3
            try (var w = new SimpleWriter(tmp)) { pqv.write(w, OnDiskGraphIndex.CURRENT_VERSION); }
            g.bytes("pqvectors_bytes", Files.readAllBytes(tmp));

            // ---- scores as raw bits: precomputed (PQDecoder, rows 2/5/6), direct (scoreFunctionFor), diversity (pair table) ----
            int[] node1 = new int[32];
            for (int p = 0; p < node1.length; p++) node1[p] = (p * 61) % N;
            g.ints("div_node1", node1);
            for (var vsf : VectorSimilarityFunction.values()) {
                float[] adc = new float[Q * N], direct = new float[Q * N], div = new float[node1.length * N];
                for (int q = 0; q < Q; q++) {
                    var pre = pqv.precomputedScoreFunctionFor(qv.get(q), vsf);
                    var dir = pqv.scoreFunctionFor(qv.get(q), vsf);
                    for (int i = 0; i < N; i++) {
                        adc[q * N + i] = pre.similarityTo(i);
                        direct[q * N + i] = dir.similarityTo(i);
                    }
                }
                for (int p = 0; p < node1.length; p++) {
                    var df = pqv.diversityFunctionFor(node1[p], vsf);
                    for (int i = 0; i < N; i++) div[p * N + i] = df.similarityTo(i);
                }
                g.floats("adc_" + vsf.name(), adc, Q, N);
                g.floats("direct_" + vsf.name(), direct, Q, N);
                g.floats("diversity_" + vsf.name(), div, node1.length, N);

                // ---- VamanaDiversityProvider.retainDiverse with the PQ build-score provider: candidates = the 48 best by the
                //      diversity function of node p (NodeArray order), selections as a byte mask ----
                var bsp = BuildScoreProvider.pqBuildScoreProvider(vsf, pqv);
                var vdp = new VamanaDiversityProvider(bsp, 1.2f);
                final int C = 48;
                int[] candIds = new int[node1.length * C];
                float[] candSc = new float[node1.length * C];
                byte[] selected = new byte[node1.length * C];
                float[] shortEdges = new float[node1.length];
                for (int p = 0; p < node1.length; p++) {
                    NodeArray na = new NodeArray(C);
                    var df = pqv.diversityFunctionFor(node1[p], vsf);
                    // top-C by (score desc, then node asc) through NodeArray.insertSorted over a strided scan of the nodes
                    NodeArray all = new NodeArray(N);
                    for (int i = 0; i < N; i++) if (i != node1[p]) all.insertSorted(i, df.similarityTo(i));
                    for (int j = 0; j < C; j++) na.addInOrder(all.getNode(j), all.getScore(j));
                    BitSet sel = new FixedBitSet(C);
                    shortEdges[p] = (float) vdp.retainDiverse(na, DEG, 0, sel);
                    for (int j = 0; j < C; j++) {
                        candIds[p * C + j] = na.getNode(j);
                        candSc[p * C + j] = na.getScore(j);
                        selected[p * C + j] = (byte) (sel.get(j) ? 1 : 0);
                    }
                }
                g.ints("rd_cand_" + vsf.name(), candIds, node1.length, C);
                g.floats("rd_scores_" + vsf.name(), candSc, node1.length, C);
                g.bytes("rd_selected_" + vsf.name(), selected, node1.length, C);
                g.floats("rd_short_edges_" + vsf.name(), shortEdges);
            }

            // ---- NVQ: NVQuantization.compute / encodeAll (deterministic: a grid search, no RNG) and NVQScorer, 1 and 3 sub-vectors ----
            for (int S : new int[]{1, 3}) {
                NVQuantization nvq = NVQuantization.compute(ravv, S);
                float[] mean = new float[D];
                for (int j = 0; j < D; j++) mean[j] = nvq.globalMean.get(j);
                g.floats("nvq_mean_s" + S, mean);
                NVQVectors nv = (NVQVectors) nvq.encodeAll(ravv);
                byte[] nb = new byte[N * D];
                float[] np = new float[N * S * 4];
                for (int i = 0; i < N; i++) {
                    var qvec = nv.get(i);
                    int off = 0;
                    for (int sv = 0; sv < S; sv++) {
                        var sub = qvec.subVectors[sv];
                        for (int d = 0; d < sub.bytes.length(); d++) nb[i * D + off + d] = sub.bytes.get(d);
                        off += sub.bytes.length();
                        np[(i * S + sv) * 4] = sub.minValue;               // the order QuantizedSubVector.write serialises them
                        np[(i * S + sv) * 4 + 1] = sub.maxValue;
                        np[(i * S + sv) * 4 + 2] = sub.growthRate;
                        np[(i * S + sv) * 4 + 3] = sub.midpoint;
                    }
                }
                g.bytes("nvq_bytes_s" + S, nb, N, D);
                g.floats("nvq_params_s" + S, np, N, S, 4);
                for (var vsf : VectorSimilarityFunction.values()) {
                    float[] sc = new float[Q * N];
                    for (int q = 0; q < Q; q++) {
                        var f = nv.scoreFunctionFor(qv.get(q), vsf);
                        for (int i = 0; i < N; i++) sc[q * N + i] = f.similarityTo(i);
                    }
                    g.floats("nvq_scores_" + vsf.name() + "_s" + S, sc, Q, N);
                }
                try (var w = new SimpleWriter(tmp)) { nv.write(w, OnDiskGraphIndex.CURRENT_VERSION); }
                g.bytes("nvqvectors_bytes_s" + S, Files.readAllBytes(tmp));
            }

            // ---- a graph built by the reference (hierarchy on), dumped level by level; searches over it with the PQ score
            //      function + exact reranker: ids, scores, the four counters ----
            var vsfBuild = VectorSimilarityFunction.COSINE;
            ImmutableGraphIndex graph;
            try (var builder = new GraphIndexBuilder(ravv, vsfBuild, DEG, BEAM, 1.2f, 1.2f, true)) {
                graph = builder.build(ravv);
            }
            var view = graph.getView();
            g.ints("graph_entry", new int[]{view.entryNode().node, view.entryNode().level, graph.getMaxLevel()});
            for (int lvl = 0; lvl <= graph.getMaxLevel(); lvl++) {
                int deg = graph.getDegree(lvl);
                List<Integer> nodes = new ArrayList<>();
                for (var it = graph.getNodes(lvl); it.hasNext(); ) nodes.add(it.nextInt());
                nodes.sort(Integer::compare);
                int[] ids = new int[nodes.size()], nb = new int[nodes.size() * deg];
                java.util.Arrays.fill(nb, -1);
                for (int r = 0; r < ids.length; r++) {
                    ids[r] = nodes.get(r);
                    int j = 0;
                    for (var it = view.getNeighborsIterator(lvl, ids[r]); it.hasNext(); ) nb[r * deg + j++] = it.nextInt();
                }
                g.ints("graph_nodes_l" + lvl, ids);
                g.ints("graph_nbrs_l" + lvl, nb, ids.length, deg);
            }
            for (var vsf : VectorSimilarityFunction.values()) {
                int[] ids = new int[Q * TOPK], counters = new int[Q * 4];
                float[] sc = new float[Q * TOPK];
                java.util.Arrays.fill(ids, -1);
                java.util.Arrays.fill(sc, Float.NEGATIVE_INFINITY);
                try (var searcher = new GraphSearcher(graph)) {
                    for (int q = 0; q < Q; q++) {
                        var ssp = new DefaultSearchScoreProvider(pqv.precomputedScoreFunctionFor(qv.get(q), vsf), ravv.rerankerFor(qv.get(q), vsf));
                        SearchResult r = searcher.search(ssp, TOPK, RERANK, 0.0f, 0.0f, Bits.ALL);
                        var ns = r.getNodes();
                        for (int j = 0; j < ns.length; j++) {
                            ids[q * TOPK + j] = ns[j].node;
                            sc[q * TOPK + j] = ns[j].score;
                        }
                        counters[q * 4] = r.getVisitedCount();
                        counters[q * 4 + 1] = r.getExpandedCount();
                        counters[q * 4 + 2] = r.getExpandedCountBaseLayer();
                        counters[q * 4 + 3] = r.getRerankedCount();
                    }
                }
                g.ints("search_ids_" + vsf.name(), ids, Q, TOPK);
                g.floats("search_scores_" + vsf.name(), sc, Q, TOPK);
                g.ints("search_counters_" + vsf.name(), counters, Q, 4);
            }

            // ---- the on-disk index: FusedPQ + inline vectors, current format version ----
            Path odgi = Files.createTempFile("golden", ".odgi");
            var wb = new OnDiskGraphIndexWriter.Builder(graph, odgi).with(new FusedPQ(graph.maxDegree(), pqv.getCompressor())).with(new InlineVectors(D));
            var suppliers = new EnumMap<FeatureId, IntFunction<Feature.State>>(FeatureId.class);
            suppliers.put(FeatureId.FUSED_PQ, ordinal -> new FusedPQ.State(view, pqv, ordinal));
            suppliers.put(FeatureId.INLINE_VECTORS, ordinal -> new InlineVectors.State(ravv.getVector(ordinal)));
            try (var writer = wb.build()) { writer.write(suppliers); }
            g.bytes("odgi_bytes", Files.readAllBytes(odgi));
            // and what the reference itself reads back out of it: fused scores for a few (origin, neighbour index) pairs
            try (var rs = new SimpleMappedReader.Supplier(odgi); var onDisk = OnDiskGraphIndex.load(rs, 0); var dv = onDisk.getView()) {
                int[] origins = new int[24];
                for (int i = 0; i < origins.length; i++) origins[i] = (i * 83) % N;
                g.ints("fused_origins", origins);
                for (var vsf : VectorSimilarityFunction.values()) {
                    float[] fs = new float[Q * origins.length * DEG];
                    java.util.Arrays.fill(fs, Float.NEGATIVE_INFINITY);
                    for (int q = 0; q < Q; q++) {
                        var f = dv.approximateScoreFunctionFor(qv.get(q), vsf);
                        for (int o = 0; o < origins.length; o++) {
                            f.enableSimilarityToNeighbors(origins[o]);
                            int j = 0;
                            for (var it = dv.getNeighborsIterator(0, origins[o]); it.hasNext(); it.nextInt(), j++)
                                fs[(q * origins.length + o) * DEG + j] = f.similarityToNeighbor(origins[o], j);
                        }
                    }
                    g.floats("fused_scores_" + vsf.name(), fs, Q, origins.length, DEG);
                }
            }
        }
        System.out.println("wrote " + target.toAbsolutePath());
    }
}
