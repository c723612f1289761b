package mclab.deploy

import java.util.concurrent.ExecutorService

import com.typesafe.config.{Config, ConfigFactory}
import mclab.gpu.Dpf
import mclab.lsh.{DpfFamilyExport, LSH}
import mclab.lsh.vector.{DenseVector, Vectors}
import mclab.utils.LocalitySensitivePartitioner

import scala.collection.mutable.ArrayBuffer
import scala.io.Source

/**
  * Drop-in for the reference's DensevectorRDFInit (src/main/scala/mclab/deploy/DensevectorRDFInit.scala): same
  * object name, same public signatures, same configuration keys, but the LSH tables live on the GPU behind
  * `mclab.gpu.Dpf` instead of in `Array[RandomDrawTreeMap]`.  Method-by-method:
  *
  *   initializeRDFHashMap  (:50-118)   -> Dpf.create + setHashFamily + setPartitioner + setSamplingIndex
  *   newFastFit / newMultiThreadFit (:127-206) -> Dpf.fit (ids = line counter, as :181,190)
  *   NewMultiThreadQueryBatch (:335-360) -> Dpf.queryBatch; overload (:372-399) fits the queries first
  *   queryBatch / querySingleKey / query (:277-317, :533-557) -> Dpf.queryBatch
  *   topKAndPrecisionScore (:472-507)  -> Dpf.queryTopK (descending inner product) + the ground-truth loop
  *   getDtAndHtNumDistribution (:515-530) -> Dpf.subindexDistribution
  *   clearAndClose (:453-458)          -> Dpf.destroy
  *
  * `queryThreadNum`, `mclab.insertThreadNum` and the thread-pool arguments are accepted and ignored: the device
  * runs every (query, table) pair of a batch at once.
  */
private[mclab] object DensevectorRDFInit {
  private var tableNum = 0
  private var permutationNum = 0
  private var partitionBits = 0
  private var dim = 0
  private var gpu: Long = 0L                      // dpf_index*; replaces vectorDatabase / vectorIdToVector (:26-28)
  private var fitted: Long = 0L                   // ids handed out so far (the size of the reference's dataTable)

  private def typeOfIndexCode(name: String): Int = name match {
    case "original" => Dpf.IndexOriginal
    case "sampling" => Dpf.IndexSampling
    case "continueBitsCount" => Dpf.IndexContinueBits
    case "angleNewMethod" => Dpf.IndexAngleNew
    case other => throw new IllegalArgumentException("unknown mclab.lsh.typeOfIndex " + other)
  }

  /** The int fields of dpf_config from the reference's configuration keys (TestSettings.scala:8-45). */
  private def cfgInts(conf: Config, numFunctions: Int, numTables: Int): Array[Int] = Dpf.cfgInts(Map(
    "family" -> (if (conf.getString("mclab.lsh.name") == "pStable") Dpf.FamilyPStable else Dpf.FamilyAngle),
    "dim" -> conf.getInt("mclab.lsh.vectorDim"),
    "num_functions" -> numFunctions,
    "num_tables" -> numTables,
    "chain_length" -> conf.getInt("mclab.lshTable.chainLength"),
    "partition_bits" -> conf.getInt("mclab.lsh.partitionBits"),
    "bucket_bits" -> conf.getInt("mclab.lshTable.bucketBits"),
    "dir_node_size" -> conf.getInt("mclab.lshTable.dirNodeSize"),
    "bucket_overflow" -> conf.getInt("mclab.lshTable.bufferOverflow"),
    "type_of_index" -> typeOfIndexCode(conf.getString("mclab.lsh.typeOfIndex")),
    "exclude_self" -> Dpf.SelfCompat,
    "metric" -> Dpf.MetricDot,                     // :490 ranks by descending inner product
    "top_k" -> conf.getInt("mclab.lsh.topK"),
    "dtype" -> Dpf.F64,                            // the JVM's doubles end to end
    "device" -> (if (conf.hasPath("mclab.gpu.device")) conf.getInt("mclab.gpu.device") else 0)))

  def initializeRDFHashMap(conf: Config): Unit = {
    tableNum = conf.getInt("mclab.lsh.tableNum")
    permutationNum = conf.getInt("mclab.lsh.permutationNum")
    partitionBits = conf.getInt("mclab.lsh.partitionBits")
    dim = conf.getInt("mclab.lsh.vectorDim")
    val numTables = tableNum * permutationNum       // :107
    val flat = DpfFamilyExport.flatten(LSHServer.lshEngine, dim)
    // PStableHashFamily.pick and the file loaders return tableNum chains only; the reference then indexes past the
    // list at LSH.scala:154 when permutationNum > 1
    if (flat.numChains != numTables)
      throw new IndexOutOfBoundsException(s"${flat.numChains} hash chains for ${numTables} tables")
    if (gpu != 0L) Dpf.destroy(gpu)
    gpu = Dpf.create(cfgInts(conf, flat.numFunctions, numTables), 0L, -1L)
    fitted = 0L
    Dpf.setHashFamily(gpu, flat.fns, flat.b, flat.w, flat.chainFn)
    if (partitionBits > 0) {
      // one independently drawn LocalitySensitivePartitioner per table, built exactly as :63-77 does
      val confForPartitioner = ConfigFactory.parseString(
        s"""
           |mclab.confType=partition
           |mclab.lsh.vectorDim=32
           |mclab.lsh.tableNum = 1
           |mclab.lshTable.chainLength=$partitionBits
           |mclab.lsh.generateMethod=default
        """.stripMargin).withFallback(conf)
      val parts = (0 until numTables).map { _ =>
        val p = new LocalitySensitivePartitioner[Int](confForPartitioner, 0, partitionBits)
        DpfFamilyExport.flattenChains(List(p.getLSH().tableIndexGenerators.head), 32)   // chain 0, Partitioner.scala:59
      }
      val pstable = parts.head.b != null
      Dpf.setPartitioner(gpu, parts.flatMap(_.fns).toArray,
        if (pstable) parts.flatMap(_.b).toArray else null,
        if (pstable) parts.flatMap(_.w).toArray else null)
    }
    if (conf.getString("mclab.lsh.typeOfIndex") == "sampling") Dpf.setSamplingIndex(gpu, DpfFamilyExport.samplingIndex())
  }

  private def readVectors(fileName: String): Array[DenseVector] = {
    val file = getClass.getClassLoader.getResource(fileName).getFile
    Source.fromFile(file).getLines().map { line =>
      val tmp = Vectors.parseDense(line)
      new DenseVector(tmp._1, tmp._2)
    }.toArray
  }

  private def fitRows(vectors: Array[DenseVector]): Array[Int] = {
    val rows = new Array[Double](vectors.length * dim)
    for (i <- vectors.indices) {
      require(vectors(i).values.length == dim, s"vector $i has ${vectors(i).values.length} of $dim values")
      Array.copy(vectors(i).values, 0, rows, i * dim, dim)
    }
    val first = fitted.toInt
    Dpf.fit(gpu, rows, vectors.length)
    fitted += vectors.length
    Array.range(first, first + vectors.length)      // ids are insertion counters (:181,190)
  }

  def newFastFit(fileName: String, conf: Config): Array[DenseVector] = newMultiThreadFit(fileName, conf)

  def newMultiThreadFit(fileName: String, conf: Config): Array[DenseVector] = {
    if (LSHServer.lshEngine == null) LSHServer.lshEngine = new LSH(conf)
    initializeRDFHashMap(conf)
    println("Finish initialize the RDF.")
    val all = readVectors(fileName)
    fitRows(all)
    println("finish load, dataTable totally has " + fitted + " objects.")
    all
  }

  private def flatQueries(vectors: Array[DenseVector], n: Int): Array[Double] = {
    val q = new Array[Double](n * dim)
    for (i <- 0 until n) Array.copy(vectors(i).values, 0, q, i * dim, dim)
    q
  }

  private def sets(queryArray: Array[Int], vectors: Array[DenseVector], steps: Int): Array[Set[AnyRef]] = {
    val n = queryArray.length
    val offsets = new Array[Long](n + 1)
    val ids =
      try {
        if (gpu == 0L) throw new NullPointerException("no index")
        Dpf.queryBatch(gpu, queryArray, flatQueries(vectors, n), n, steps, offsets)
      } catch {
        case _: NullPointerException => println("need to fit the data first"); java.util.Arrays.fill(offsets, 0L); Array.empty[Int]
      }
    Array.tabulate(n)(i => ids.slice(offsets(i).toInt, offsets(i + 1).toInt).map(x => Int.box(x): AnyRef).toSet)
  }

  /** NewMultiThreadQueryBatch without moving the Sets: they stay on the device.  Returns their sizes; fetchSets then
    * brings over the Sets of a range of the batch's queries, topKOfResidentSets reranks all of them (the order of
    * topKAndPrecisionScore).  A batch's Sets can be gigabytes of ids, which a caller that looks at a few of them, or only
    * at the top-k, never needs on the JVM heap.  Valid until the next query / fit call on this object. */
  def queryBatchSizes(queryArray: Array[Int], denseVectorArray: Array[DenseVector], steps: Int = 0): Array[Long] = {
    val n = queryArray.length
    residentOffsets = new Array[Long](n + 1)
    if (gpu == 0L) { println("need to fit the data first"); return new Array[Long](n) }
    try Dpf.queryBatchResident(gpu, queryArray, flatQueries(denseVectorArray, n), n, steps, residentOffsets)
    catch { case _: NullPointerException => println("need to fit the data first"); java.util.Arrays.fill(residentOffsets, 0L) }
    Array.tabulate(n)(i => residentOffsets(i + 1) - residentOffsets(i))
  }

  def fetchSets(q0: Int, q1: Int): Array[Set[AnyRef]] = {
    val base = residentOffsets(q0)
    val ids = Dpf.setsFetch(gpu, q0, q1, residentOffsets(q1) - base)
    Array.tabulate(q1 - q0)(i => ids.slice((residentOffsets(q0 + i) - base).toInt, (residentOffsets(q0 + i + 1) - base).toInt)
      .map(x => Int.box(x): AnyRef).toSet)
  }

  def topKOfResidentSets(k: Int): (Array[Array[Int]], Array[Array[Double]]) = {
    val n = residentOffsets.length - 1
    val ids = new Array[Int](n * k)
    val scores = new Array[Double](n * k)
    Dpf.setsTopK(gpu, n, k, ids, scores)
    (Array.tabulate(n)(i => ids.slice(i * k, (i + 1) * k).filter(_ >= 0)), Array.tabulate(n)(i => scores.slice(i * k, (i + 1) * k)))
  }

  private var residentOffsets: Array[Long] = Array(0L)

  def NewMultiThreadQueryBatch(queryArray: Array[Int], denseVectorArray: Array[DenseVector], steps: Int = 0,
                               queryThreadNum: Int = 5): Array[Set[AnyRef]] = sets(queryArray, denseVectorArray, steps)

  /** The "system use" overload (:372-399): the queries are put into the index first and queried under their new ids. */
  def NewMultiThreadQueryBatch(querySVArray: Array[DenseVector], steps: Int, queryThreadNum: Int): Array[Set[AnyRef]] = {
    val queryArray = fitRows(querySVArray)
    sets(queryArray, querySVArray, steps)
  }

  def queryBatch(queryArray: Array[Int], denseVectorArray: Array[DenseVector], steps: Int, L: Int): Array[Set[AnyRef]] = {
    require(L == tableNum * permutationNum, "a prefix of the tables needs a handle created with table_end = L")
    sets(queryArray, denseVectorArray, steps)
  }

  def querySingleKey(queryKey: Int, denseVector: DenseVector, steps: Int = 0, L: Int = tableNum * permutationNum): Set[AnyRef] =
    queryBatch(Array(queryKey), Array(denseVector), steps, L)(0)

  def query(queryKeyArray: Array[Int], querySVArray: Array[DenseVector], steps: Int = 0, queryThreadNum: Int = 10,
            queryThreadPool: ExecutorService): Array[Set[AnyRef]] = {
    if (queryThreadPool != null) queryThreadPool.shutdown()        // the reference shuts the pool down after its tasks (:549)
    sets(queryKeyArray, querySVArray, steps)
  }

  def getTopKGroundTruth(filename: String, K: Int): Array[Set[Int]] = {
    val gtFile = getClass.getClassLoader.getResource(filename).getFile
    Source.fromFile(gtFile).getLines().map(line => Vectors.analysisKNN(line, K).toSet).toArray
  }

  def clearAndClose(): Unit = {
    if (gpu != 0L) Dpf.destroy(gpu)
    gpu = 0L
    fitted = 0L
  }

  def topKAndPrecisionScore(allDenseVectors: Array[DenseVector], groundTruth: Array[Set[Int]], conf: Config, steps: Int = 0,
                            queryThreadPool: ExecutorService): (Array[Array[Int]], Double) = {
    val n = groundTruth.length
    val k = conf.getInt("mclab.lsh.topK")
    val queryArray = groundTruth.indices.toArray
    val ids = new Array[Int](n * k)
    val scores = new Array[Double](n * k)
    Dpf.queryTopK(gpu, queryArray, flatQueries(allDenseVectors, n), n, steps, k, ids, scores)
    if (queryThreadPool != null) queryThreadPool.shutdown()
    var averageScore = 0.0
    val allQueryedTopK = new ArrayBuffer[Array[Int]]
    for (i <- 0 until n) {
      val top = ids.slice(i * k, (i + 1) * k).filter(_ >= 0)      // -1 pads a candidate set smaller than k
      if (top.nonEmpty) {                                         // an empty candidate set contributes no row (:479)
        allQueryedTopK += top
        averageScore += top.count(groundTruth(i).contains).toDouble / n
      }
    }
    (allQueryedTopK.toArray, averageScore / k)
  }

  def getDtAndHtNumDistribution(): (Array[Double], Array[Double]) = {
    val ht = new Array[Double](1 << partitionBits)
    if (gpu != 0L) Dpf.subindexDistribution(gpu, ht)
    // the data table is a flat array on the device: one "sub-index" holding every vector
    (Array(fitted.toDouble), ht)
  }
}
