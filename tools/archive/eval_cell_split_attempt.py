import re
L = open('/tmp/eval_cell_old.txt').read().split('\n')
def rng(a, b): return L[a-1:b]
def ind(lines, n=2): return [(' ' * n + l) if (l.strip() and not l.startswith('#')) else l for l in lines]
def lam2fn(lines, header):
    """a top-level lambda `  auto name = [&](...) ... {` ... `  };` -> member function with `header` as its first line"""
    out = list(lines)
    out[0] = header
    assert out[-1].strip() == '};', out[-1]
    out[-1] = '  }'
    return out
FI = '  __device__ __forceinline__ '
out = []
A = lambda s: out.append(s if isinstance(s, str) else '\n'.join(s))
A(open('/tmp/eval_head.txt').read())
A(rng(1, 2))
A('struct CellEval {')
A('  static_assert(!PRESET || (REPAIR_INLINE && LAT == 0 && !RES), "PRESET: k_repair\'s instantiation");')
A('  static_assert(LAT == 0 || (!STRICT && !EXT), "the latency form exists for FAST math launches of <= kMaxBatch poses (DBG: phase stamps only)");')
A('''  // ---- what the kernel hands over
  const EvalParams &P;
  const SlotArgs &SA;
  const int cl, pose_idx;
  unsigned char *const smem;
  const ResCell rc;
  const uint2 preset;
  __device__ __forceinline__ CellEval(const EvalParams &P_, const SlotArgs &SA_, int cl_, int pose_idx_, unsigned char *smem_, const ResCell rc_, const uint2 preset_)
      : P(P_), SA(SA_), cl(cl_), pose_idx(pose_idx_), smem(smem_), rc(rc_), preset(preset_) {}

  // ---- LDS layout and per-cell scalars (members are initialised in this order)''')
A([l.replace('  constexpr int NC = eval_hist_copies(NT, NB);', '  static constexpr int NC = eval_hist_copies(NT, NB);') for l in rng(7, 50)])
A(rng(68, 75))
A(rng(104, 105))
A('  static constexpr bool kResRezero = RES && NT <= 256 && NID_XPOSE_SUM;')
A(rng(107, 110))
A('  static constexpr bool kLatEarly = LAT > 0 && !RES;')
A('  static constexpr int kLatN = LAT > 0 ? LAT : 1;')
A(rng(113, 116))
A(rng(154, 155))
A('  static constexpr int tiny_scaled_hi = (int)((1023u - 512u - (unsigned)NID_TINY_W_EXP) << 20);')
A(rng(157, 160))
A(rng(443, 450))
A(rng(451, 459))
A(rng(461, 463))
A('  static constexpr bool use_lane_masks = !STRICT && LAT == 0 && !DBG && !BIG;')
A('  static constexpr bool use_gomask = JAC && use_lane_masks;')
A('''  LatPix lat[LAT > 0 ? LAT : 1];  // LAT + JAC: the cost phase's hand-over to the Jacobian phase
  // ---- what the fold leaves for the phases behind it
  uint4 flagw;                 // clamp_flag as read behind the cost phase: clamped | near-saturated | lin_flag[2]
  bool any_sat, any_clamped, any_ns;
  double cw[4] = {0.0, 0.0, 0.0, 0.0};  // the clamped samples' four target weights, as the sample path computes them
  double nw[4] = {0.0, 0.0, 0.0, 0.0};  // ... and the near-saturated samples'
  int jc_cl = 0, jc_ns = 0;
  unsigned grp_colz = 0u;  // grp_flag (wave-uniform)
  unsigned lin_col1, lin_colz;
  double Hc, Hj;               // the cell's entropies, in scalar registers
  double err, rho0, rho1;      // residual and Huber weights (residual_and_huber: wave 0, in a tail)
  double acc[6];               // the six Jacobian sums: this thread's, then the cell's
''')
# zero_histograms
A(rng(77, 78))
A(lam2fn(rng(79, 103), FI + 'void zero_histograms(int first) {'))
A('')
A('  // level-1 edge: never evaluated (computeH.cu:271-275).  true: published as inactive, the workgroup is done')
A(FI + 'bool head() {')
A(ind([l.replace('return false;', 'return true;') for l in rng(51, 66)]))
A('    return false;\n  }\n')
A(FI + 'void setup() {')
A('    NID_STAMP(0);')
A(ind(rng(118, 145)))
A(ind(rng(161, 171)))
A('  }\n')
A(rng(146, 153))
A(lam2fn(rng(172, 433), '  template <class PrescaledTag, class RepairTag>\n' + FI + 'void hist_add(int jr, int jc, const double (&wr_in)[4], const double (&win)[4], PrescaledTag prescaled, double pcv, int group, RepairTag repair_tag) {'))
A(lam2fn(rng(434, 442), FI + 'void dump_pixel(int s, const PixelFront &f, double ic, int jc, const double (&wc)[4]) {'))
A(rng(469, 469))
A(lam2fn(rng(470, 494), '  template <class RepairTag>\n' + FI + 'void strict_cost_loop(RepairTag repair_tag) {'))
A(rng(495, 499))
A(lam2fn(rng(500, 584), '  template <class SecondTag>\n' + FI + 'bool cost_round(int sb, int r, const TileIn &cur, TileIn &nxt, SecondTag second_pass) {'))
A(rng(585, 586))
A(lam2fn(rng(587, 613), FI + 'void repair_round(int sb) {'))
A(rng(614, 614))
A(lam2fn(rng(615, 629), '  template <class Body>\n' + FI + 'void second_pass_rounds(unsigned todo, Body &&body) {'))
A('')
A('  // ---- phase 1: cost -------------------------------------------------------------------------------------------')
A(FI + 'void cost_phase() {')
A(ind(rng(630, 728)))
A('  }\n')
A(rng(781, 782))
A(lam2fn(rng(783, 834), '  template <class CoarseTag>\n' + FI + 'unsigned long long fold_bin(int b, CoarseTag coarse_tag, double &mass_out) {'))
A(rng(835, 835))
A(lam2fn(rng(836, 840), FI + 'int lin_word(int b, int &bit) {'))
A(rng(869, 872))
A(lam2fn(rng(873, 883), FI + 'bool repair_wanted() {  // (the fold\'s verdict, from LDS: wave-uniform)'))
A(rng(884, 885))
A(lam2fn(rng(886, 889), FI + 'bool done() {'))
A(lam2fn(rng(890, 910), FI + 'bool deferred_to_repair() {  // wave 0, in a tail'))
A('')
A('  // ---- fold the copies, probabilities, entropies, weight tables ------------------------------------------------')
A(FI + 'void fold() {')
fold = rng(731, 780)
fold = [l.replace('  uint4 flagw = *reinterpret_cast', '  flagw = *reinterpret_cast')
         .replace('  const bool any_sat = !STRICT', '  any_sat = !STRICT')
         .replace('  const bool any_clamped = any_sat && flagw.x != 0u, any_ns = any_sat && flagw.y != 0u;', '  any_clamped = any_sat && flagw.x != 0u; any_ns = any_sat && flagw.y != 0u;')
         .replace('  const unsigned lin_col1 = ', '  lin_col1 = ').replace('  const unsigned lin_colz = ', '  lin_colz = ') for l in fold]
drop = ('  double cw[4] = {0.0, 0.0, 0.0, 0.0};', '  double nw[4] = {0.0, 0.0, 0.0, 0.0};', '  int jc_cl = 0, jc_ns = 0;', '  unsigned grp_colz = 0u;')
fold = [l for l in fold if not l.startswith(drop)]
A(ind(fold))
A(ind(rng(841, 868)))
A(ind(rng(911, 953)))
A('    Hc = wave_uniform(0.0 - ent[0]);')
A('    Hj = wave_uniform(0.0 - ent[1]);')
A('  }\n')
A(rng(956, 956))
A(lam2fn(rng(958, 969), FI + 'void residual_and_huber() {'))
A('')
A('  // the tail of a cost-only evaluation; returns what eval_cell returns')
A(FI + 'bool cost_only_tail() {')
co = rng(971, 1002)   # inside `if (!JAC) { ... }`
A(co)
A('  }\n')
A('  // ---- phase 2: Jacobian (recompute, see header comment) --------------------------------------------------------')
A(FI + 'void jacobian_phase() {')
A(ind(rng(1005, 1010)))
A(ind(rng(1013, 1037)))
jp = rng(1039, 1324)
A(ind(jp))
A('  }\n')
A(rng(1325, 1327))
A('  // true: this wave retires (its return value is done()); false: wave 0 carries on into the tail with the cell\'s six sums')
A(FI + 'bool block_sum() {')
bs = rng(1328, 1383)
bs = [l.replace('return done();', 'return true;') for l in bs]
A(ind(bs))
A('    return false;\n  }\n')
A('  // the cell\'s Jacobian, quadratic form and hand-off (wave 0); returns what eval_cell returns')
A(FI + 'bool jacobian_tail() {')
A(ind(rng(1384, 1435)))
A('  }\n')
A('''  __device__ __forceinline__ bool run() {
    if (head()) return false;
    setup();
    cost_phase();
    fold();
    if (!JAC) return cost_only_tail();
    jacobian_phase();
    if (block_sum()) return done();
    return jacobian_tail();
  }
};

// One cell at one pose by one workgroup: the evaluation kernels' body.  RES: called from the resident kernel (k_resident) -- the
// B-spline table is in LDS already, the cell's count and reference entropy are in registers, and the histograms were zeroed
// behind the previous request.  Returns true for a cell that wants the repair pass and publishes nothing (loop-form kernels: it
// has been queued for k_repair).''')
A(rng(1, 2))
A('''__device__ __forceinline__ bool eval_cell(const EvalParams &P, const SlotArgs &SA, const int cl, const int pose_idx, unsigned char *smem,
                                          const ResCell rc = ResCell{0, 0.0, true}, const uint2 preset = uint2{0u, 0u}) {
  CellEval<NT, JAC, STRICT, NB, DBG, EXT, LAT, BIG, RES, REPAIR_INLINE, PRESET> c(P, SA, cl, pose_idx, smem, rc, preset);
  return c.run();
}

}  // namespace nid''')
open('/tmp/nid_eval_cell.hip.h', 'w').write('\n'.join(out) + '\n')
print("ok", sum(s.count('\n') + 1 for s in out))

# ---- second attempt (later in round 6): the two arrays that kept the whole object in memory.  fold_bin's select chains on the members
# cw[4] / nw[4] reach SROA as one load at a selected offset into the object; as locals of fold() they are separate small allocas again
# (profiles/r06_ablations_A.txt item 10: no scratch any more, +-2..5 VGPRs per kernel, +3.4 % on the headline kernel: reverted again).
p = '/tmp/nid_eval_cell.hip.h'
s = open(p).read()
members = """  double cw[4] = {0.0, 0.0, 0.0, 0.0};  // the clamped samples' four target weights, as the sample path computes them
  double nw[4] = {0.0, 0.0, 0.0, 0.0};  // ... and the near-saturated samples'
  int jc_cl = 0, jc_ns = 0;
"""
assert members in s
s = s.replace(members, '')
s = s.replace('unsigned long long fold_bin(int b, CoarseTag coarse_tag, double &mass_out) {',
              'unsigned long long fold_bin(int b, CoarseTag coarse_tag, double &mass_out, const double (&cw)[4], const double (&nw)[4], int jc_cl, int jc_ns) {')
s = s.replace('fold_bin(b, std::true_type{}, mass)', 'fold_bin(b, std::true_type{}, mass, cw, nw, jc_cl, jc_ns)')
s = s.replace('fold_bin(b, std::false_type{}, mass)', 'fold_bin(b, std::false_type{}, mass, cw, nw, jc_cl, jc_ns)')
s = s.replace("""    any_clamped = any_sat && flagw.x != 0u; any_ns = any_sat && flagw.y != 0u;
    if (any_sat) {""", """    any_clamped = any_sat && flagw.x != 0u; any_ns = any_sat && flagw.y != 0u;
""" + members.replace('\n  ', '\n    ').replace('  double cw', '    double cw', 1) + """    if (any_sat) {""")
open(p, 'w').write(s)

