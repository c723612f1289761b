package mclab.lsh

import mclab.lsh.hashFamilies.{AngleParameterSet, LSHTableHashChain, PStableParameterSet, Sampling}

/**
  * Flattens the hash chains an `LSH` instance holds into the arrays `mclab.gpu.Dpf.setHashFamily` /
  * `setPartitioner` take.  Lives in package mclab.lsh because `LSH.tableIndexGenerators`,
  * `LSHTableHashChain.chainedHashFunctions` and the parameter-set classes are package-private
  * (LSH.scala:27, LSHTableHashChain.scala:12-14, AngleHashFamily.scala:11, PStableHashFamily.scala:191).
  *
  * The pool handed to the device is simply every chain's functions in order, so
  * chainFn(t)(j) = t * chainLength + j.  (Chains that share functions - the permutationNum shuffles of
  * AngleHashFamily.scala:143-146 - may be de-duplicated by the caller; the result is the same.)
  */
object DpfFamilyExport {
  /** fns [F*dim], b [F] (null for angle), w [F] (null for angle), chainFn [L*chainLength], F. */
  case class Flat(fns: Array[Double], b: Array[Double], w: Array[Int], chainFn: Array[Int], numFunctions: Int,
                  numChains: Int, chainLength: Int)

  /** One function's coefficients as a dense row.  The dense path of the reference multiplies the stored
    * `values` positionally (SimilarityCalculator.scala:46), so a function must carry all `dim` values. */
  private def denseRow(values: Array[Double], indices: Array[Int], dim: Int): Array[Double] = {
    require(values.length == dim && indices.indices.forall(i => indices(i) == i),
      s"hash function stores ${values.length} of $dim coefficients: the dense query path would mis-align them")
    values
  }

  def flatten(lsh: LSH, dim: Int): Flat = flattenChains(lsh.tableIndexGenerators, dim)

  def flattenChains(chains: List[LSHTableHashChain[_]], dim: Int): Flat = {
    val m = chains.head.chainLength
    require(chains.forall(_.chainLength == m), "chains of different lengths")
    val params = chains.flatMap(_.chainedHashFunctions)
    val pstable = params.head.isInstanceOf[PStableParameterSet]
    val fns = new Array[Double](params.length * dim)
    val b = if (pstable) new Array[Double](params.length) else null
    val w = if (pstable) new Array[Int](params.length) else null
    for ((p, f) <- params.zipWithIndex) p match {
      case a: AngleParameterSet =>
        Array.copy(denseRow(a.oneFunction.values, a.oneFunction.indices, dim), 0, fns, f * dim, dim)
      case s: PStableParameterSet =>
        Array.copy(denseRow(s.a.values, s.a.indices, dim), 0, fns, f * dim, dim)
        b(f) = s.b
        w(f) = s.w
    }
    Flat(fns, b, w, Array.range(0, params.length), params.length, chains.length, m)
  }

  /** The bit permutation `typeOfIndex = sampling` applies (Sampling.scala:6-11, seeded as LSH.scala:21). */
  def samplingIndex(seed: Int = 88387): Array[Int] = new Sampling(seed).samplingIndex.toArray
}
