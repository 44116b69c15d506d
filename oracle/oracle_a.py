"""Oracle A -- literal CPU restatement of the reference's term algebra.

TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product path
(cfd_amd/) never does.

Parity status: the reference (Haskell, GHC 7.8.3) cannot be built or run in
this environment, so this file follows the reference *source text*.  The only
reference-owned known answers for this path are the two helper-level values
pinned in tests (solveUnknown testEquation, distributeMultiply testTerms);
for the sweep itself parity is UNPINNED by the reference.

Structure mirrors the reference one function at a time (same names, same
fold orders, same operand orders) so that a reader can diff it by eye:

  Term / Expression / fmap                src/SolutionDomain.hs:89-103
  getSign, chooseSlope, minmod, superbee  src/SolutionDomain.hs:105-107, 869-897
  offsetPosition (clamp at domain edge)   src/SolutionDomain.hs:434-448
  offsetPosition_adj, getNode, getNodeVal src/SolutionDomain.hs:450-468
  average                                 src/SolutionDomain.hs:475-479
  approximateDerivative_adj               src/SolutionDomain.hs:506-538
  prop_adj (incl. Speed, Vorticity)       src/SolutionDomain.hs:540-575
  propLimitedSlope_adj                    src/SolutionDomain.hs:644-659
  propCentralDiff_adj                     src/SolutionDomain.hs:687-713
  sideArea_adj, sideLength_adj            src/SolutionDomain.hs:840-867
  solveUnknown, multTerm, integ*          src/CalculusUtils.hs:17-19, 36-118
  d_ ... ddfm_, divergence ... paired*    src/CalculusUtils.hs:120-267
  equations, calcSteps_adj                src/Main.hs:89-187
  applyDiffEq_adj, correctPrevTime,
  runSingleStep, totalDensity             src/Main.hs:189-270
  makeRows, adjGraphToGrid                src/Main.hs:334-346

Slow by design (closures rebuilt per cell, per equation, per sweep exactly as
the reference does); meant for grids of a few hundred cells.

Arithmetic: Python floats are IEEE binary64; `**` on Double in GHC is libm
pow, here math.pow.  Division by zero follows IEEE (inf/nan) as in Haskell.
"""
import math

# ---------------------------------------------------------------- enums
# data Side = Now | Prev | East | West | North | South | Top | Bottom | Center
NOW, PREV, EAST, WEST, NORTH, SOUTH, TOP, BOTTOM, CENTER = range(9)
SIDES_FROM_EAST = [EAST, WEST, NORTH, SOUTH, TOP, BOTTOM, CENTER]   # enumFrom East
# data Direction = Time | X | Y | Z
TIME, X, Y, Z = range(4)
DIRS_FROM_X = [X, Y, Z]                                             # enumFrom X
# data Property = Speed | Vorticity | Pressure | U | V | W | Density | Temperature | Mew | Dye
SPEED, VORTICITY, PRESSURE, U, V, W, DENSITY, TEMPERATURE, MEW, DYE = range(10)
TEMPORAL, SPATIAL = 0, 1
DIRECTIONAL, NONDIRECTIONAL = 0, 1


def sideIndex(s):                      # src/SolutionDomain.hs:63-73
    return {EAST: 0, WEST: 1, NORTH: 2, SOUTH: 3, TOP: 4, BOTTOM: 5, CENTER: 6, NOW: 7, PREV: 8}[s]


def propIndex(p):                      # src/SolutionDomain.hs:77-85
    return {U: 0, V: 1, W: 2, DENSITY: 3, TEMPERATURE: 4, MEW: 5, DYE: 6}[p]


def dirIndex(d):                       # src/SolutionDomain.hs:153-158
    return {X: 0, Y: 1, Z: 2}[d]


# ---------------------------------------------------------------- IEEE helpers
def fdiv(a, b):
    """IEEE-754 binary64 division (Python raises on a zero divisor; Haskell does not)."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return float("nan")
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -math.inf if neg else math.inf


def fpow(a, b):
    """libm pow with IEEE results instead of Python exceptions."""
    try:
        return math.pow(a, b)
    except (ValueError, OverflowError):
        if a < 0:
            return float("nan")
        return math.inf


# ---------------------------------------------------------------- physical constants
storedSteps = 3                                   # src/SolutionDomain.hs:133-134
timeStep = 0.00001                                # :237-238
initialMew = 0.1                                  # :240-241
initialTemperature = 223.0                        # :243-244
sutherlandConstant = 120.0                        # :246-247
sutherlandLambda = fdiv(initialMew * (initialTemperature + sutherlandConstant),
                        fpow(initialTemperature, 1.5))          # :249-251
specificHeatCp = 1005.0                           # :256-257
specificHeatCv = 716.0                            # :259-260
gasConstantR = specificHeatCp - specificHeatCv    # :253-254
heatConductivityK = 0.025                         # src/Main.hs:55-56


# ---------------------------------------------------------------- terms
class Constant:
    __slots__ = ("val",)

    def __init__(self, val):
        self.val = val


class Unknown:
    __slots__ = ("coeff",)

    def __init__(self, coeff):
        self.coeff = coeff


class SubExpression:
    __slots__ = ("terms",)

    def __init__(self, terms):
        self.terms = list(terms)


class Derivative_adj:
    __slots__ = ("denom", "function", "centered", "multiplier")

    def __init__(self, denom, function, centered, multiplier):
        self.denom = denom            # Direction
        self.function = function      # SchemeType -> AdjPos -> Side -> Term
        self.centered = centered      # Side
        self.multiplier = multiplier  # SchemeType -> AdjPos -> Side -> Double


def fmap(f, x):                        # src/SolutionDomain.hs:99-103
    if isinstance(x, Constant):
        return Constant(f(x.val))
    if isinstance(x, Unknown):
        return Unknown(f(x.coeff))
    raise ValueError("can't fmap other than unknown and constant")


def val(t):
    """Record selector `val`; partial in the reference too (Constant only)."""
    if isinstance(t, Constant):
        return t.val
    raise ValueError("No match in record selector val")


class Equation:
    def __init__(self, lhs, rhs, unknownProperty):
        self.lhs, self.rhs, self.unknownProperty = lhs, rhs, unknownProperty


# ---------------------------------------------------------------- limiter
POSITIVE, ZERO, NEGATIVE = range(3)


def getSign(d):                        # src/SolutionDomain.hs:105-107
    if d == 0:
        return ZERO
    return POSITIVE if d > 0 else NEGATIVE


def hmin(x, y):                        # GHC Ord Double: min x y = if x <= y then x else y
    return x if x <= y else y


def hmax(x, y):                        # max x y = if x <= y then y else x
    return y if x <= y else x


def chooseSlopeHelper(f1, f2, x, y):   # :869-879 ; returns None for Nothing
    sign = getSign(x)
    if getSign(x) == getSign(y):
        if sign == POSITIVE:
            return f1(x, y)
        if sign == NEGATIVE:
            return f2(x, y)
        return None
    return None


def chooseSlope(f1, f2, n):            # :881-888 ; foldM in Maybe starting at head n
    acc = n[0]
    for y in n:
        acc = chooseSlopeHelper(f1, f2, acc, y)
        if acc is None:
            return 0.0
    return acc


def minmod(n):                         # :890-891
    return chooseSlope(hmin, hmax, n)


def maxmod(n):                         # :893-894
    return chooseSlope(hmax, hmin, n)


def superbee(a, b):                    # :896-897
    return minmod([maxmod([a, b]), minmod([2 * a, 2 * b])])


# ---------------------------------------------------------------- state
class AdjNode:
    __slots__ = ("faceArea", "edgeLen", "property", "neighbors", "origPos", "active", "index")

    def __init__(self, faceArea, edgeLen, prop, neighbors, origPos, active, index):
        self.faceArea, self.edgeLen, self.property = faceArea, edgeLen, prop
        self.neighbors, self.origPos, self.active, self.index = neighbors, origPos, active, index


class AdjGraph:
    def __init__(self, grids, currModTime, currAbsTime):
        self.grids = grids                # list[storedSteps] of list[AdjNode]
        self.currModTime = currModTime
        self.currAbsTime = currAbsTime


def pushBackTime(t):                   # :422-423
    return (t - 1) % storedSteps


def pushUpTime(t):                     # :425-426
    return (t + 1) % storedSteps


def isUpperSide(side):                 # :402-408
    return side in (EAST, NORTH, TOP, NOW)


def directionFromCenter(side):         # :410-420
    return {EAST: X, WEST: X, NORTH: Y, SOUTH: Y, TOP: Z, BOTTOM: Z, NOW: TIME, PREV: TIME}[side]


def boundaryPair(d):                   # :172-177
    return {X: (EAST, WEST), Y: (NORTH, SOUTH), Z: (TOP, BOTTOM), TIME: (NOW, PREV)}[d]


def faceToDirections(s):               # :179-189
    return {EAST: [Y, Z], WEST: [Y, Z], NORTH: [X, Z], SOUTH: [X, Z],
            TOP: [X, Y], BOTTOM: [X, Y], CENTER: [X, Y, Z]}[s]


def offsetPosition(pos, side, maxPos):
    """src/SolutionDomain.hs:434-448 on a spatial coordinate tuple (timePos is 0 in every node)."""
    if side in (CENTER, NOW, PREV):
        return pos
    upper = isUpperSide(side)
    direction = directionFromCenter(side)
    i = dirIndex(direction)
    boundary = maxPos[i] if upper else 0
    moved = pos[i] + (1 if upper else -1)
    moved = min(boundary, moved) if upper else max(boundary, moved)
    p = list(pos)
    p[i] = moved
    return tuple(p)


def makePositions(maxes):              # :388-392 ; first coordinate slowest
    coords = [()]
    for m in maxes:
        coords = [c + (k,) for c in coords for k in range(m + 1)]
    return coords


def build_graph(maxPos, gridSize, fields, active, cellLength=None):
    """Equivalent of initialGrid_adj (src/SolutionDomain.hs:308-353) for caller-supplied
    values: `fields` is 7 sequences in propIndex order (U,V,W,Density,Temperature,Mew,Dye),
    `active` a sequence of bools, both in node order (x slowest, z fastest).
    Geometry follows cellLength (:150-151) and the avMap/slMap folds (:279-286); a caller that
    already holds the three cell lengths passes them as `cellLength` (gridSize is then unused)."""
    if cellLength is None:
        cellLength = [fdiv(gridSize[i], float(maxPos[i] + 1)) for i in range(3)]
    slMap = {d: cellLength[dirIndex(d)] for d in DIRS_FROM_X}
    avMap = {}
    for s in SIDES_FROM_EAST:
        a = 1.0
        for d in faceToDirections(s):
            a = a * slMap[d]
        avMap[s] = a
    positions = makePositions(maxPos)
    locations = {p: i for i, p in enumerate(positions)}
    nodes = []
    for i, p in enumerate(positions):
        faceArea = [None] * 7
        for s, a in avMap.items():
            faceArea[sideIndex(s)] = a
        edgeLen = [slMap[X], slMap[Y], slMap[Z]]
        prop = [float(fields[k][i]) for k in range(7)]
        neighbors = [None] * 7
        for s in SIDES_FROM_EAST:
            neighbors[sideIndex(s)] = locations[offsetPosition(p, s, maxPos)]
        nodes.append(AdjNode(faceArea, edgeLen, prop, neighbors, p, bool(active[i]), i))
    # V.replicate storedSteps: every slot starts as the same node list (:352)
    grids = [[_copy_node(n) for n in nodes] for _ in range(storedSteps)]
    return AdjGraph(grids, 0, 0)


def _copy_node(n):
    return AdjNode(n.faceArea, n.edgeLen, list(n.property), n.neighbors, n.origPos, n.active, n.index)


def getNode(g, x):                     # :450-451
    i, t = x
    return g.grids[t][i]


def getNodeVal(g, x, p):               # :454-457
    return getNode(g, x).property[propIndex(p)]


def offsetPosition_adj(x, side, env):  # :459-468
    idx, t = x
    if side in (CENTER, NOW):
        return (idx, t)
    if side == PREV:
        return (idx, pushBackTime(t))
    n = getNode(env, (idx, t)).neighbors[sideIndex(side)]
    return (n, t) if n is not None else (idx, t)


def average(terms):                    # :475-479
    ln = float(len(terms))
    p = 0.0
    for n in terms:
        p = p + fdiv(n, ln)
    return p


def sideArea_adj(side, x, env):        # :840-850
    if side in (NOW, PREV):
        return 1.0
    return getNode(env, x).faceArea[sideIndex(side)]


def sideLength_adj(direction, x, env):  # :858-867
    if direction == TIME:
        return timeStep
    return getNode(env, x).edgeLen[dirIndex(direction)]


def propCentralDiff_adj(prp, x, side, env):   # :687-713
    idx, t = x
    neigh = offsetPosition_adj((idx, t), side, env)
    neighborPos = getNode(env, neigh).origPos
    originalPos = getNode(env, (idx, t)).origPos

    def getVal(p):
        v = getNodeVal(env, p, prp)
        if v is None:
            raise ValueError("no value")
        return v
    if originalPos == neighborPos:
        return getVal((idx, t))
    return average([getVal((idx, t)), getVal(neigh)])


def propLimitedSlope_adj(prp, position, side, env):   # :644-659
    valCenter = propCentralDiff_adj(prp, position, CENTER, env)
    d = directionFromCenter(side)
    upper, lower = boundaryPair(d)
    upperNVal, lowerNVal = [propCentralDiff_adj(prp, offsetPosition_adj(position, s, env), CENTER, env)
                            for s in (upper, lower)]
    ave = superbee(upperNVal - valCenter, valCenter - lowerNVal)
    if isUpperSide(side):
        return valCenter + 0.5 * ave
    return valCenter - 0.5 * ave


def prop_adj(schemeType, prp, pos, side, env):    # :543-575
    def scheme(p, ps, sd, e):
        if sd == CENTER or sd == NOW or sd == PREV:
            return propCentralDiff_adj(p, ps, sd, e)
        return propLimitedSlope_adj(p, ps, sd, e)     # Directional and Nondirectional are identical
    if prp == MEW:
        t = scheme(TEMPERATURE, pos, side, env)
        return fdiv(sutherlandLambda * fpow(t, 1.5), t + sutherlandConstant)
    if prp == PRESSURE:
        return gasConstantR * scheme(TEMPERATURE, pos, side, env) * scheme(DENSITY, pos, side, env)
    momentums = [U, V, W]                         # enumFrom U \\ enumFrom Density
    if prp == SPEED:
        return magnitude([scheme(nxt, pos, side, env) for nxt in momentums])
    if prp == VORTICITY:
        pairs = [(x, y) for i, x in enumerate(momentums) for y in momentums[i + 1:]]

        def component(a, b):
            def makeDeriv(m, n):
                return Derivative_adj(directionFromConvection(m),
                                      lambda _st, p, s, n=n: Constant(scheme(n, p, s, env)),
                                      EAST,                                  # "east is arbitrary here"
                                      lambda _st, _p, _s: 1.0)
            deriv1 = makeDeriv(a, b)
            deriv2 = makeDeriv(b, a)
            return val(approximateDerivative_adj(deriv1, pos, env)) - val(approximateDerivative_adj(deriv2, pos, env))
        return magnitude([component(a, b) for a, b in pairs])
    return scheme(prp, pos, side, env)


def magnitude(v):                      # :540-541
    prev = 0.0
    for nVal in v:
        prev = prev + (nVal * nVal)
    return math.sqrt(prev)


def directionFromConvection(d):        # :127-131
    return {U: X, V: Y, W: Z}[d]


def approximateDerivative_adj(deriv, position, vs):   # :506-538
    direction, func, side, m = deriv.denom, deriv.function, deriv.centered, deriv.multiplier
    if directionFromCenter(side) == direction:
        neighbor = offsetPosition_adj(position, side, vs)
        sl = sideLength_adj(direction, position, vs)
        sln = sideLength_adj(direction, neighbor, vs)
        interval = average([sl, sln])
        thisVal = func(NONDIRECTIONAL, position, CENTER)
        neighborVal = func(NONDIRECTIONAL, neighbor, CENTER)
        neighborIsUpper = isUpperSide(side)

        def f(first):
            return neighborVal if first == neighborIsUpper else thisVal
        mult = m(NONDIRECTIONAL, position, CENTER)
        c1, c2 = val(f(True)), val(f(False))
        return Constant(fdiv((c1 - c2) * mult, interval))
    s1, s2 = boundaryPair(direction)
    n1 = offsetPosition_adj(position, s1, vs)
    n2 = offsetPosition_adj(position, s2, vs)
    sl = sideLength_adj(direction, position, vs)
    sln1 = sideLength_adj(direction, n1, vs)
    sln2 = sideLength_adj(direction, n2, vs)
    interval = 2 * average([sl, sln1, sln2])
    n1Val = func(NONDIRECTIONAL, n1, side)
    n2Val = func(NONDIRECTIONAL, n2, side)
    mult = m(NONDIRECTIONAL, position, side)
    c1, c2 = val(n1Val), val(n2Val)
    return Constant(fdiv((c1 - c2) * mult, interval))


# ---------------------------------------------------------------- CalculusUtils
def direcDimenType(direc):             # src/CalculusUtils.hs:8-11
    return TEMPORAL if direc == TIME else SPATIAL


def volumeOrInterval_adj(dimetype, x, env):   # :17-19
    if dimetype == TEMPORAL:
        return timeStep
    p = 1.0
    for d in DIRS_FROM_X:
        p = p * sideLength_adj(d, x, env)
    return p


def d2C(d):                            # :28-33
    return {X: U, Y: V, Z: W}[d]


def solveUnknown(eq, position, vs):    # :36-52
    def sumExpression(s, terms):
        acc = 0.0
        for t in terms:
            acc = s(acc, t)
        return acc

    def sumUnknown(p, n):
        if isinstance(n, Unknown):
            return p + n.coeff
        if isinstance(n, SubExpression):
            return p + sumExpression(sumUnknown, n.terms)
        return p + 0.0

    def sumConstants(p, n):
        if isinstance(n, Constant):
            return p + n.val
        if isinstance(n, Derivative_adj):
            return p + sumExpression(sumConstants, [approximateDerivative_adj(n, position, vs)])
        if isinstance(n, SubExpression):
            return p + sumExpression(sumConstants, n.terms)
        return p + 0.0
    lhsUnknown = sumExpression(sumUnknown, eq.lhs)
    rhsUnknown = sumExpression(sumUnknown, eq.rhs)
    lhsConstants = sumExpression(sumConstants, eq.lhs)
    rhsConstants = sumExpression(sumConstants, eq.rhs)
    return fdiv(rhsConstants - lhsConstants, lhsUnknown - rhsUnknown)


def distributeMultiply(terms, m):      # :54-55
    out = []
    for t in terms:
        out.extend(multTerm(m, t))
    return out


def multTerm(m, term):                 # :58-66
    if isinstance(term, SubExpression):
        return distributeMultiply(term.terms, m)
    if isinstance(term, Derivative_adj):
        func = term.function

        def modf(st, x, s, func=func, m=m):
            return fmap(lambda v: v * m, func(st, x, s))
        return [Derivative_adj(term.denom, modf, term.centered, term.multiplier)]
    return [fmap(lambda v: v * m, term)]


def integSurface_adj(f, position, direction, unknownProp, vs):   # :68-91
    s1, s2 = boundaryPair(direction)

    def value(s, isUpper):
        if isUpper:
            term = f(s)
        else:
            (term,) = multTerm(-1.0, f(s))
        sideAreaVal = sideArea_adj(s, position, vs)
        isUnknown = (direcDimenType(direction), isUpper) == (TEMPORAL, True)
        if isinstance(term, Derivative_adj):
            return multTerm(sideAreaVal, Derivative_adj(term.denom, term.function, s, term.multiplier))[0]
        if isinstance(term, SubExpression):
            return SubExpression(distributeMultiply(term.terms, sideAreaVal))
        if isUnknown:
            k = fdiv(sideAreaVal, prop_adj(DIRECTIONAL, unknownProp, position, s, vs))
            constantVal = fmap(lambda v: v * k, term)
            return Unknown(1.0 if math.isnan(val(constantVal)) else val(constantVal))
        return fmap(lambda v: v * sideAreaVal, term)
    return [value(s1, True), value(s2, False)]


def integSingleTerm_adj(term, dimetype, cellposition, unknownProp, vs):   # :93-110
    def nonDerivAnswer():
        if isinstance(term, SubExpression):
            raise ValueError("can't integrate a subexpression as a single term")
        return multTerm(volumeOrInterval_adj(dimetype, cellposition, vs), term)
    if isinstance(term, Derivative_adj):
        if direcDimenType(term.denom) == dimetype:
            func = term.function
            return integSurface_adj(lambda s: func(DIRECTIONAL, cellposition, s),
                                    cellposition, term.denom, unknownProp, vs)
        return nonDerivAnswer()
    return nonDerivAnswer()


def integ_adj(vs, dimetype, terms, cellposition, unknownProp):   # :112-118
    out = []
    for x in terms:
        out.extend(integSingleTerm_adj(x, dimetype, cellposition, unknownProp, vs))
    return out


def integUnknown(env, unknownProp):    # :260-267, curried as the equations use it
    def integrate(dimetype, terms, cellposition):
        return integ_adj(env, dimetype, terms, cellposition, unknownProp)
    return integrate


def multProps_adj(props, schemeType, env):   # :231-239 ; left product starting from 1
    def prod(pos, side):
        acc = 1.0
        for nxt in props:
            acc = acc * prop_adj(schemeType, nxt, pos, side, env)
        return acc
    return prod


def dfm_(properties, factor, multF, direction, env):   # :127-132
    def func(st, x, s):
        return Constant(multProps_adj(properties, st, env)(x, s) * factor)
    return Derivative_adj(direction, func, CENTER, multF)


def _one(_st, _x, _s):
    return 1.0


def df_(properties, factor, direction, env):   # :124-125
    return dfm_(properties, factor, _one, direction, env)


def d_(properties, direction, env):            # :120-121
    return df_(properties, 1.0, direction, env)


def ddfm_(inner, factor, multF, direction):    # :140-146 ; `inner` is an already-built Term
    def func(_st, _x, _s):
        return multTerm(factor, inner)[0]
    return Derivative_adj(direction, func, CENTER, multF)


def ddf_(inner, factor, direction):            # :137-138
    return ddfm_(inner, factor, _one, direction)


def dd_(inner, direction):                     # :134-135
    return ddf_(inner, 1.0, direction)


def divergence(vector):                        # :201-204
    return [dd_(t, d) for t, d in zip(vector, DIRS_FROM_X)]


def gradient(properties, constantFactor, env):  # :206-207
    return [df_(properties, constantFactor, d, env) for d in DIRS_FROM_X]


def divergenceWithPropsFactor(properties, constantFactor, env):   # :217-219
    return [df_(list(properties) + [d2C(x)], constantFactor, x, env) for x in DIRS_FROM_X]


def divergenceWithProps(properties, env):      # :214-215
    return divergenceWithPropsFactor(properties, 1.0, env)


def divGrad(properties, constantFactor, env):  # :221-224
    return divergence(gradient(properties, constantFactor, env))


def squareDerivative(properties, constantFactor, direction, env):   # :241-247
    foldedProps_st = lambda st, pos, side: multProps_adj(properties, st, env)(pos, side)
    return [ddf_(d_(list(properties) + list(properties), direction, env), fdiv(constantFactor, 2.0), direction),
            ddfm_(d_(properties, direction, env), constantFactor * (-1.0), foldedProps_st, direction)]


def pairedMultipliedDerivatives(props1, props2, dir1, dir2, env):   # :249-258
    p1 = lambda st, pos, side: multProps_adj(props1, st, env)(pos, side)
    p2 = lambda st, pos, side: multProps_adj(props2, st, env)(pos, side)
    return [dd_(d_(list(props1) + list(props2), dir1, env), dir2),
            ddfm_(d_(props1, dir1, env), -1.0, p2, dir2),
            ddfm_(d_(props2, dir1, env), -1.0, p1, dir2)]


def integrateTerms(integrate, terms):          # :209-212 + integrateOrder :226-229
    def mk(term):
        return lambda pos: integrate(TEMPORAL, integrate(SPATIAL, [term], pos), pos)
    return [mk(t) for t in terms]


def _time_then_space(integrate, term):
    return lambda pos: integrate(SPATIAL, integrate(TEMPORAL, [term], pos), pos)


# ---------------------------------------------------------------- equations (src/Main.hs:89-175)
def continuity(env):
    integrate = integUnknown(env, DENSITY)
    return Equation(
        [_time_then_space(integrate, d_([DENSITY], TIME, env))]
        + integrateTerms(integrate, divergenceWithProps([DENSITY], env)),
        [lambda pos: [Constant(0.0)]],
        DENSITY)


def _momentum(env, comp, pdir):
    """uMomentum / vMomentum / wMomentum (src/Main.hs:98-140) differ only in the component and direction."""
    integrate = integUnknown(env, comp)
    dp = d_([PRESSURE], pdir, env)
    lhs = ([_time_then_space(integrate, d_([DENSITY, comp], TIME, env))]
           + integrateTerms(integrate, divergenceWithProps([DENSITY, comp], env))
           + [lambda pos: integrate(TEMPORAL, integrate(SPATIAL, [dp], pos), pos)])
    rhs = []
    for group in (divGrad([MEW, comp], 1.0, env),
                  divergence([d_([MEW, U], pdir, env), d_([MEW, V], pdir, env), d_([MEW, W], pdir, env)]),
                  [ddf_(d, -(2.0 / 3.0), pdir) for d in divergenceWithProps([MEW], env)]):
        rhs += integrateTerms(integrate, group)
    return Equation(lhs, rhs, comp)


def uMomentum(env):
    return _momentum(env, U, X)


def vMomentum(env):
    return _momentum(env, V, Y)


def wMomentum(env):
    return _momentum(env, W, Z)


def dye(env):
    integrate = integUnknown(env, DYE)
    lhs = ([_time_then_space(integrate, d_([DENSITY, DYE], TIME, env))]
           + integrateTerms(integrate, divergenceWithProps([DENSITY, DYE], env)))
    rhs = integrateTerms(integrate, divGrad([MEW, DYE], 0.001, env))
    return Equation(lhs, rhs, DYE)


def energy(env):
    integrate = integUnknown(env, TEMPERATURE)
    lhs = ([_time_then_space(integrate, df_([DENSITY, TEMPERATURE], specificHeatCv, TIME, env))]
           + integrateTerms(integrate, divergenceWithPropsFactor([DENSITY, TEMPERATURE], specificHeatCv, env))
           + integrateTerms(integrate, divergenceWithProps([PRESSURE], env)))
    rhs = []
    for group in (divGrad([TEMPERATURE], heatConductivityK, env),
                  squareDerivative([MEW, U], 8.0 / 3.0, X, env),
                  squareDerivative([MEW, V], 8.0 / 3.0, Y, env),
                  squareDerivative([MEW, W], 8.0 / 3.0, Z, env),
                  squareDerivative([MEW, U], 1.0, Y, env),
                  squareDerivative([MEW, V], 1.0, X, env),
                  squareDerivative([MEW, U], 1.0, Z, env),
                  squareDerivative([MEW, W], 1.0, X, env),
                  squareDerivative([MEW, V], 1.0, Z, env),
                  squareDerivative([MEW, W], 1.0, Y, env),
                  pairedMultipliedDerivatives([MEW, U], [V], X, Y, env),
                  pairedMultipliedDerivatives([MEW, U], [W], X, Z, env),
                  pairedMultipliedDerivatives([MEW, W], [V], Z, Y, env)):
        rhs += integrateTerms(integrate, group)
    return Equation(lhs, rhs, TEMPERATURE)


def getDiscEqInstance(eq, pos):        # src/Main.hs:177-178
    l, r = [], []
    for t in eq.lhs:
        l.extend(t(pos))
    for t in eq.rhs:
        r.extend(t(pos))
    return Equation(l, r, eq.unknownProperty)


calcSteps_adj = [(continuity, True), (uMomentum, False), (vMomentum, False),
                 (wMomentum, False), (energy, True), (dye, False)]      # src/Main.hs:180-187


def setNodeProperty(node, prp, value):  # src/Main.hs:189-190
    p = list(node.property)
    p[propIndex(prp)] = value
    return AdjNode(node.faceArea, node.edgeLen, p, node.neighbors, node.origPos, node.active, node.index)


def applyDiffEq_adj(equations, stepTime, g):   # src/Main.hs:202-226
    cmt, cat = g.currModTime, g.currAbsTime
    new_cmt = pushUpTime(cmt) if stepTime else cmt
    new_cat = cat + 1 if stepTime else cat
    source_grid = g.grids[cmt]
    env = g
    new_aln = []
    for i in range(storedSteps):
        grid = g.grids[i]
        if i == new_cmt:
            out = []
            for node in source_grid:
                pos = (node.index, cmt)
                prev = node
                for next_eq, alwaysRun in equations:
                    if alwaysRun or prev.active:
                        discEquation = getDiscEqInstance(next_eq(env), pos)
                        newValue = solveUnknown(discEquation, pos, env)
                        prev = setNodeProperty(prev, discEquation.unknownProperty, newValue)
                out.append(prev)
            new_aln.append(out)
        else:
            new_aln.append(grid)
    return AdjGraph(new_aln, new_cmt, new_cat)


def correctPrevTime(g, equations):     # src/Main.hs:228-258
    cmt, cat = g.currModTime, g.currAbsTime
    if not cat > 1:
        return g
    to_be_corrected_t = pushBackTime(cmt)
    concrete_t = pushBackTime(to_be_corrected_t)
    prev_grid = g.grids[to_be_corrected_t]
    env = g
    corrected = []
    for prev_node in prev_grid:
        j = prev_node.index
        prev = prev_node
        for next_eq, alwaysRun in equations:
            solvedProperty = next_eq(env).unknownProperty
            c = prop_adj(NONDIRECTIONAL, solvedProperty, (j, cmt), CENTER, env)
            pp = prop_adj(NONDIRECTIONAL, solvedProperty, (j, concrete_t), CENTER, env)
            newValue = pp + (0.5 * (c - pp))
            if alwaysRun or prev.active:
                prev = setNodeProperty(prev, solvedProperty, newValue)
        corrected.append(prev)
    new_aln = [corrected if i == to_be_corrected_t else g.grids[i] for i in range(storedSteps)]
    return AdjGraph(new_aln, cmt, cat)


def runSingleStep(g):                  # src/Main.hs:192-200
    prev = g
    for nxt in (0, 1, 2):
        stepTime = nxt == 0
        predicted = applyDiffEq_adj(calcSteps_adj, stepTime, prev)
        prev = correctPrevTime(predicted, calcSteps_adj)
    return prev


def totalDensity(env):                 # src/Main.hs:269-270
    t = pushBackTime(env.currModTime)
    p = 0.0
    for x in range(len(env.grids[t])):
        p = p + prop_adj(NONDIRECTIONAL, DENSITY, (x, t), CENTER, env)
    return p


# ---------------------------------------------------------------- output slices (src/Main.hs:334-352)
def allPositionsCurrentTime(env):      # src/Main.hs:262-263
    return [(idx, env.currModTime) for idx in range(len(env.grids[env.currModTime]))]


def makeRows(whole, curr, items, r, width):    # src/Main.hs:334-337 ; clause order matters: [] is tested first
    items = list(items)
    i = 0
    while True:
        if i == len(items):
            return whole + [curr]
        if r == 0:
            whole, curr, r = whole + [curr], [], width
            continue
        curr = curr + [items[i]]
        i += 1
        r -= 1


def adjGraphToGrid(vs, prp, rowLength, zLevel):    # src/Main.hs:339-346
    picked = [c for c in allPositionsCurrentTime(vs) if getNode(vs, c).origPos[2] == zLevel]
    return makeRows([], [], [prop_adj(NONDIRECTIONAL, prp, c, PREV, vs) for c in picked], rowLength, rowLength)


# ---------------------------------------------------------------- known-answer inputs (src/Main.hs:79-85, :309)
def addTerm(terms, term):              # src/Main.hs:70-71
    return terms + [term]


def addTerms(terms1, terms2):          # src/Main.hs:73-77
    for x in terms2:
        terms1 = addTerm(terms1, x)
    return terms1


def testEquation():
    return Equation(
        [Unknown(2.0), SubExpression([Constant(2.0), Constant(2.0), Unknown(0.025)]), Constant(-0.232)],
        addTerms([Unknown(-0.025), Unknown(-0.05)], [Constant(2.0), Constant(3.0)]),
        U)


def testTerms():
    return [Unknown(2.4), Constant(1.2), Constant(3.112), Unknown(-0.21),
            SubExpression([Constant(2.0), Constant(2.0), SubExpression([Unknown(0.33333)])])]


# ---------------------------------------------------------------- array-level conveniences for tests
def slot_fields(g, slot_rel=0):
    """7 lists (propIndex order) of slot cmt-slot_rel."""
    t = (g.currModTime - slot_rel) % storedSteps
    return [[n.property[k] for n in g.grids[t]] for k in range(7)]
