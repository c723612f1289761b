package mclab.gpu

/**
  * The native side of the GPU batch-query engine: one-to-one with jni/dpf_jni.c, which is one-to-one with
  * the C ABI of include/dpf.h.  A handle (Long) is a `dpf_index*`; it serves one call at a time.
  */
object Dpf {
  System.loadLibrary("dpf_jni")

  // enum values of include/dpf.h
  val FamilyAngle = 0
  val FamilyPStable = 1
  val IndexOriginal = 0
  val IndexSampling = 1
  val IndexContinueBits = 2
  val IndexAngleNew = 3
  val SelfCompat = 0          // reproduces RandomDrawTreeMap.java:982 (boxed-Integer reference inequality)
  val MetricDot = 0           // DensevectorRDFInit.scala:490
  val MetricL2 = 1
  val ProbeFlips = 0          // RandomDrawTreeMap.getSimilarWithStepWiseFaster (RandomDrawTreeMap.java:742-797)
  val ProbeOwn = 1            // RandomDrawTreeMap.getSimilarWithStepWise(key, steps) (RandomDrawTreeMap.java:630-675)
  val F32 = 0
  val F64 = 1
  // DPF_OPT_* of dpf_set_option
  val OptEmitCap = 1
  val OptBucketListCap = 2
  val OptSyncSizes = 3
  val OptFirstPassPermille = 4
  val OptSeedSample = 5
  val OptShortRerank = 6
  val OptProjectionMinRows = 7
  val OptScanPasses = 8
  val OptEarlyPassPermille = 9
  val OptExactSeeds = 10
  val OptKeepForwarded = 11
  val OptSideStream = 12
  val OptProbeLanes = 13
  val OptDeferBuild = 14
  val OptExpectedRows = 15
  val OptListBins = 16
  // DPF_FLAG_* of dpf_config.flags
  val FlagHashOnly = 1
  val FlagGatherOnly = 2
  val FlagHostForest = 4
  val FlagFullFilter = 8
  val FlagNoInt8 = 16
  val FlagNoCentre = 32
  val FlagBucketOrderRows = 64
  val FlagNoBucketOrderRows = 128

  /** The int32 fields of `dpf_config`, in declaration order (checked against include/dpf.h by tests/test_jni_shim.py). */
  val ConfigIntFields: Array[String] = Array(
    "family", "dim", "num_functions", "num_tables", "chain_length", "partition_bits", "bucket_bits", "dir_node_size",
    "bucket_overflow", "type_of_index", "exclude_self", "metric", "top_k", "dtype", "device", "flags", "table_begin",
    "table_end", "subindex_group", "subindex_groups")

  /** The array `create` takes, from named values; a missing name is 0 (flags, table shard = all tables). */
  def cfgInts(values: Map[String, Int]): Array[Int] = {
    val unknown = values.keySet -- ConfigIntFields
    require(unknown.isEmpty, s"unknown dpf_config fields: $unknown")
    ConfigIntFields.map(name => values.getOrElse(name, 0))
  }

  @native def create(cfgInts: Array[Int], rowBegin: Long, rowEnd: Long): Long
  @native def destroy(h: Long): Unit
  @native def lastError(h: Long): String
  @native def setHashFamily(h: Long, fns: Array[Double], b: Array[Double], w: Array[Int], chainFn: Array[Int]): Unit
  @native def setPartitioner(h: Long, partFns: Array[Double], partB: Array[Double], partW: Array[Int]): Unit
  @native def setSamplingIndex(h: Long, idx: Array[Int]): Unit
  @native def fit(h: Long, rows: Array[Double], n: Long): Unit
  @native def fitFloat(h: Long, rows: Array[Float], n: Long): Unit
  @native def hashBatch(h: Long, q: Array[Double], nq: Long, hashes: Array[Int], parts: Array[Int]): Unit
  @native def queryBatch(h: Long, qids: Array[Int], q: Array[Double], nq: Long, steps: Int, offsets: Array[Long]): Array[Int]
  /** the Sets stay on the device; offsets (nq + 1) receives their sizes as a running sum */
  @native def queryBatchResident(h: Long, qids: Array[Int], q: Array[Double], nq: Long, steps: Int, offsets: Array[Long]): Unit
  /** ids of the resident Sets of the queries [q0, q1); count = offsets(q1) - offsets(q0) */
  @native def setsFetch(h: Long, q0: Long, q1: Long, count: Long): Array[Int]
  /** top-k rerank (DensevectorRDFInit.topKAndPrecisionScore's order) of every resident Set */
  @native def setsTopK(h: Long, nq: Long, k: Int, ids: Array[Int], scores: Array[Double]): Unit
  @native def queryTopK(h: Long, qids: Array[Int], q: Array[Double], nq: Long, steps: Int, k: Int, ids: Array[Int], scores: Array[Double]): Unit
  @native def rerank(h: Long, q: Array[Double], nq: Long, candOffsets: Array[Long], candIds: Array[Int], k: Int, ids: Array[Int], scores: Array[Double]): Unit
  @native def subindexDistribution(h: Long, out: Array[Double]): Unit
  @native def counters(h: Long, out: Array[Long]): Unit
  @native def setProbeMode(h: Long, mode: Int): Unit
  @native def save(h: Long, path: String): Unit
  @native def load(path: String, device: Int): Long
  /** shard = Array(table_begin, table_end, subindex_group, subindex_groups); rows [rowBegin, rowEnd), rowEnd < 0 = to the end */
  @native def loadShard(path: String, device: Int, shard: Array[Int], rowBegin: Long, rowEnd: Long): Long
  @native def setOption(h: Long, option: Int, value: Long): Unit
}
