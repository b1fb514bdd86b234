/// Host-side glue for the batched engine of libgpu_compute.so: what a maintainer of the reference adds
/// next to src/GPU.Compute.fs / src/GPU.Wrappers.fs to run a whole parallel-tempering job in ONE native
/// call (replacing the per-iteration loop of GPUTempering.optimize, src/GPU.Wrappers.fs:271-344, and
/// the dispatch of src/GPU.Wrappers.fs:696-753), and, in module TerrainBatched below, to run
/// Terrain.Generation.generate (src/Terrain.Generation.fs:172-199) for a whole batch of chunks on the voxel-chunk
/// engine.  (TerrainBatched uses Voxel and Terrain types: it belongs in a file of Terrain.fsproj, after
/// Terrain.Generation.fs; it is kept here so that the glue is in one place.)  NOT compiled in this repository's image (no .NET SDK
/// there); the C ABI it binds is include/procgen_pt.h and is exercised through the same entry points
/// by procgen_amd/native.py.
namespace GPU

open System
open System.Runtime.InteropServices
open System.Numerics
open ParallelTempering.Core

module Batched =

    [<Struct; StructLayout(LayoutKind.Sequential)>]
    type PtGraphConfig =
        val mutable struct_size: int
        val mutable node_count: int
        val mutable edge_capacity: int
        val mutable temp_count: int
        val mutable group_count: int
        val mutable swap_frequency: int      // <= 0: max 5 (int (sqrt n * 2.0))
        val mutable mutation_mode: int       // 0 = Mutations.adaptive, 1 = perturbPosition only
        val mutable rng_mode: int            // 0 = Philox keyed by SHA-256(seed)
        [<MarshalAs(UnmanagedType.ByValArray, SizeConst = 6)>]
        val mutable weights: float[]         // connectivity, endpoint, edgeLength, collision, distribution, transitions
        val mutable target_edge_length: float
        val mutable min_distance: float
        val mutable min_distance_eps: float
        val mutable philox_key: uint64
        val mutable seed: nativeint          // UTF-8 char*, may be 0
        val mutable device: int
        val mutable group_base: int
        val mutable ladder_temp_count: int
        val mutable ladder_slot_base: int
        val mutable block_threads: int
        val mutable flags: int               // 0 = defaults (incremental evaluation; placement by size)
        val mutable cluster_size: int        // 0 = choose
        val mutable reserved: int

    [<Struct; StructLayout(LayoutKind.Sequential)>]
    type PtConvergenceMetrics =
        val mutable GelmanRubin: float
        val mutable EffectiveSampleSize: float
        val mutable CostStabilization: float
        val mutable SwapAcceptanceRate: float
        val mutable HasGelmanRubin: int
        val mutable HasEss: int
        val mutable HasConverged: int
        val mutable Iteration: int

    [<Literal>]
    let private Lib = "libgpu_compute.so"

    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern void pt_graph_default_config(PtGraphConfig& cfg, int nodeCount, int edgeCapacity, int tempCount, int groupCount)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_graph_create(PtGraphConfig& cfg, nativeint& engine)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern void pt_graph_destroy(nativeint engine)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_graph_upload(nativeint engine, float32[] positions, int[] edges, int edgeStride, int[] edgeCounts,
                               byte[] nodeKind, float[] nodeValue, float[] temperatures)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_graph_run(nativeint engine, int64 steps)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_graph_run_until_converged(nativeint engine, int64 maxIterations, int checkFreq, float strictness,
                                            int64& stepsDone, [<Out>] PtConvergenceMetrics[] perLadder)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_graph_download(nativeint engine, float32[] positions, int[] edges, int edgeStride, int[] edgeCounts, float[] costs)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern nativeint pt_last_error_string()

    let private check (rc: int) =
        if rc <> 0 then
            failwithf "libgpu_compute: %s (code %d)" (Marshal.PtrToStringUTF8(pt_last_error_string())) rc

    /// NodeType of the procedural graphs (src/ProceduralGeneration.fs:9-15) as the engine's kind / value pair.
    let private kindAndValue (toKind: 'T -> byte * float) (nodes: 'T array) =
        let kinds = Array.zeroCreate<byte> nodes.Length
        let values = Array.zeroCreate<float> nodes.Length
        nodes |> Array.iteri (fun i n -> let (k, v) = toKind n in kinds.[i] <- k; values.[i] <- v)
        kinds, values

    /// Upload one ladder (chain i at temperatures.[i]), run `body` on the engine, download the chains.
    let private withLadder (temperatures: float array) (graphs: SpatialGraph<'T> array) (toKind: 'T -> byte * float)
                           (adaptive: bool) (body: nativeint -> 'R) : (Chain<'T> array * 'R) =
        let n = graphs.[0].Nodes.Length
        let t = temperatures.Length
        let maxEdges = graphs |> Array.map (fun g -> g.Edges.Length) |> Array.max
        let cap = max (2 * maxEdges) (maxEdges + 64)
        let mutable cfg = PtGraphConfig()
        pt_graph_default_config(&cfg, n, cap, t, 1)
        cfg.mutation_mode <- if adaptive then 0 else 1
        let mutable engine = 0n
        check (pt_graph_create(&cfg, &engine))
        try
            let positions = Array.zeroCreate<float32> (t * n * 3)
            let edges = Array.zeroCreate<int> (t * cap * 2)
            let counts = Array.zeroCreate<int> t
            graphs |> Array.iteri (fun c g ->
                g.Positions |> Array.iteri (fun i (p: Vector3) ->
                    positions.[(c * n + i) * 3] <- p.X
                    positions.[(c * n + i) * 3 + 1] <- p.Y
                    positions.[(c * n + i) * 3 + 2] <- p.Z)
                // list order: the head of Edges (the newest edge) first, as the reference indexes removals
                g.Edges |> List.iteri (fun k (a, b, _) ->
                    edges.[(c * cap + k) * 2] <- a
                    edges.[(c * cap + k) * 2 + 1] <- b)
                counts.[c] <- g.Edges.Length)
            let kinds, values = kindAndValue toKind graphs.[0].Nodes
            check (pt_graph_upload(engine, positions, edges, cap, counts, kinds, values, temperatures))
            let result = body engine
            let costs = Array.zeroCreate<float> t
            check (pt_graph_download(engine, positions, edges, cap, counts, costs))
            let chains =
                graphs |> Array.mapi (fun c g ->
                    let pos = Array.init n (fun i ->
                        Vector3(positions.[(c * n + i) * 3], positions.[(c * n + i) * 3 + 1], positions.[(c * n + i) * 3 + 2]))
                    let es = List.init counts.[c] (fun k -> (edges.[(c * cap + k) * 2], edges.[(c * cap + k) * 2 + 1], 1.0))
                    { State = { g with Positions = pos; Edges = es }
                      Cost = costs.[c]                      // the engine's own full evaluation of the state it returns
                      Temperature = temperatures.[c]
                      AcceptanceRate = 0.5 })
            chains, result
        finally
            pt_graph_destroy engine

    /// Replaces the GPU branch of Dispatch.optimize (src/GPU.Wrappers.fs:696-724): `iterations` steps of
    /// ParallelTempering.run (src/ParallelTempering.Core.fs:145-177) in one native call.
    let optimize (temperatures: float array) (graphs: SpatialGraph<'T> array) (toKind: 'T -> byte * float)
                 (iterations: int) : Chain<'T> array =
        withLadder temperatures graphs toKind true (fun e -> check (pt_graph_run(e, int64 iterations))) |> fst

    /// Replaces optimizeUntilConvergence (src/GPU.Wrappers.fs:728-753): Convergence.runUntilConvergence
    /// (src/ParallelTempering.Core.fs:496-550) with the history and the checks on the device.
    let optimizeUntilConvergence (temperatures: float array) (graphs: SpatialGraph<'T> array) (toKind: 'T -> byte * float)
                                 (maxIterations: int) (strictness: float)
                                 : (Chain<'T> array * Convergence.ConvergenceMetrics) =
        let checkFreq = max 50 (maxIterations / 20)
        let chains, (m: PtConvergenceMetrics) =
            withLadder temperatures graphs toKind true (fun e ->
                let out = Array.zeroCreate<PtConvergenceMetrics> 1
                let mutable steps = 0L
                check (pt_graph_run_until_converged(e, int64 maxIterations, checkFreq, strictness, &steps, out))
                out.[0])
        chains,
        { Convergence.GelmanRubin = (if m.HasGelmanRubin <> 0 then Some m.GelmanRubin else None)
          Convergence.EffectiveSampleSize = (if m.HasEss <> 0 then Some m.EffectiveSampleSize else None)
          Convergence.CostStabilization = m.CostStabilization
          Convergence.SwapAcceptanceRate = m.SwapAcceptanceRate
          Convergence.HasConverged = (m.HasConverged <> 0)
          Convergence.Iteration = m.Iteration }


/// Terrain.Generation.generate (src/Terrain.Generation.fs:172-199) for a batch of chunks in one engine: every chunk's
/// ladder (geometricLadder 0.1 5.0 6, swap every 50) runs on the GPU, the chunk seed hashCoordinates seed x y z keys its
/// random streams, and the lowest-cost replica of every ladder comes back (ParallelTempering.getBest, :196).
/// The native side is driven in this order (integration/abi_harness.c replays exactly this sequence from C and
/// tests/test_abi_harness.py compares what it prints with the CPU restatement of the terrain path):
///   pt_chunk_default_config -> pt_chunk_create -> pt_chunk_upload -> pt_chunk_set_positions -> pt_chunk_run
///   -> pt_chunk_download -> pt_chunk_destroy
module TerrainBatched =

    open Voxel
    open Terrain

    [<Struct; StructLayout(LayoutKind.Sequential)>]
    type PtChunkConfig =
        val mutable struct_size: int
        val mutable temp_count: int
        val mutable group_count: int
        val mutable swap_frequency: int       // <= 0: 50
        val mutable rng_mode: int
        val mutable device: int
        val mutable group_base: int
        val mutable flags: int
        [<MarshalAs(UnmanagedType.ByValArray, SizeConst = 8)>]
        val mutable weights: float[]          // traversability, geological, overhang, erosion, aesthetics, caves, performance, biomeTransition
        val mutable density_mutation_scale: float32
        val mutable solid_to_air_threshold: float32
        val mutable refinement_probability: float
        val mutable philox_key: uint64        // the generation seed: chunk keys are hashCoordinates of it
        val mutable seed: nativeint           // UTF-8 char*; 0 here (the key is given as a number)

    [<Literal>]
    let private Lib = "libgpu_compute.so"

    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern void pt_chunk_default_config(PtChunkConfig& cfg, int tempCount, int groupCount)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_chunk_create(PtChunkConfig& cfg, nativeint& engine)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern void pt_chunk_destroy(nativeint engine)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_chunk_upload(nativeint engine, float32[] density, float32[] moisture, float32[] temperature,
                               byte[] material, float[] temperatures)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_chunk_set_positions(nativeint engine, int[] xyz)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_chunk_run(nativeint engine, int64 steps)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern int pt_chunk_download(nativeint engine, float32[] density, float32[] moisture, float32[] temperature, float[] costs)
    [<DllImport(Lib, CallingConvention = CallingConvention.Cdecl)>]
    extern nativeint pt_last_error_string()

    let private check (rc: int) =
        if rc <> 0 then
            failwithf "libgpu_compute: %s (code %d)" (Marshal.PtrToStringUTF8(pt_last_error_string())) rc

    /// `initial`: the chunks as Generation.initializeFromNoise leaves them (src/Terrain.Generation.fs:85-120), one per
    /// position; the noise stays on the host.  Returns the optimised chunks in the same order.
    let generateFrom (initial: VoxelChunk array) (seed: uint64) (iterations: int) : VoxelChunk array =
        let pt = Generation.getPTConfig ()
        let mc = Generation.getMutationConfig ()
        let g = initial.Length
        let t = pt.TemperatureCount
        let v = ChunkSize.resolution * ChunkSize.resolution * ChunkSize.resolution
        let mutable cfg = PtChunkConfig()
        pt_chunk_default_config(&cfg, t, g)
        cfg.swap_frequency <- pt.SwapFrequency
        cfg.density_mutation_scale <- mc.DensityMutationScale
        cfg.solid_to_air_threshold <- mc.SolidToAirThreshold
        cfg.refinement_probability <- mc.RefinementProbability
        cfg.philox_key <- seed
        let mutable engine = 0n
        check (pt_chunk_create(&cfg, &engine))
        try
            // every temperature of a ladder starts from the same chunk (src/Terrain.Generation.fs:181-182)
            let density = Array.zeroCreate<float32> (g * t * v)
            let moisture = Array.zeroCreate<float32> (g * t * v)
            let temperature = Array.zeroCreate<float32> (g * t * v)
            let material = Array.zeroCreate<byte> (g * v)
            let xyz = Array.zeroCreate<int> (g * 3)
            initial |> Array.iteri (fun c chunk ->
                chunk.Graph.Nodes |> Array.iteri (fun i (n: VoxelData) ->
                    material.[c * v + i] <- n.Material
                    for k in 0 .. t - 1 do
                        let at = (c * t + k) * v + i
                        density.[at] <- n.Density
                        moisture.[at] <- n.Moisture
                        temperature.[at] <- n.Temperature)
                xyz.[c * 3] <- chunk.Position.X
                xyz.[c * 3 + 1] <- chunk.Position.Y
                xyz.[c * 3 + 2] <- chunk.Position.Z)
            let ladder = ParallelTempering.geometricLadder pt.MinTemperature pt.MaxTemperature t
            let temps = Array.init (g * t) (fun k -> ladder.[k % t])
            check (pt_chunk_upload(engine, density, moisture, temperature, material, temps))
            check (pt_chunk_set_positions(engine, xyz))
            check (pt_chunk_run(engine, int64 iterations))
            let costs = Array.zeroCreate<float> (g * t)
            check (pt_chunk_download(engine, density, moisture, temperature, costs))
            initial |> Array.mapi (fun c chunk ->
                // ParallelTempering.getBest (src/ParallelTempering.Core.fs:218-221): the first chain of lowest cost
                let best = [| 0 .. t - 1 |] |> Array.minBy (fun k -> costs.[c * t + k])
                let nodes =
                    chunk.Graph.Nodes |> Array.mapi (fun i n ->
                        let at = (c * t + best) * v + i
                        { n with Density = density.[at]; Moisture = moisture.[at]; Temperature = temperature.[at] })
                { chunk with Graph = { chunk.Graph with Nodes = nodes } })
        finally
            pt_chunk_destroy engine

    /// Drop-in for a loop of Generation.generate calls (src/Terrain.Generation.fs:172-199) over `positions`.
    let generate (positions: ChunkPosition array) (seed: uint64) (iterations: int) : VoxelChunk array =
        let initial =
            positions |> Array.map (fun pos ->
                let chunkSeed = Deterministic.DeterministicGeneration.hashCoordinates seed pos.X pos.Y pos.Z
                Generation.initializeFromNoise pos chunkSeed Noise.Presets.mountainous)
        generateFrom initial seed iterations
