{-# LANGUAGE ForeignFunctionInterface #-}
-- | Binding of libcfdhip.so for the reference program, and the marshalling of its AdjGraph.
--
-- STATUS: source only.  No GHC exists in the environment this library was built in, so this
-- module has never been compiled; it is written against the reference's types
-- (SolutionDomain.AdjGraph / AdjNode / Grids) and include/cfd_hip.h.
--
-- Use: in Main.hs replace the body of runTimeSteps_Print's fold with CfdHip.runTimeStepsHip
-- (the replacement is a one-line change of main).  The solve loop, the printing and initialGrid_adj stay in Haskell.
module CfdHip
    ( HipGraph, withHipGraph, hipRunSingleStep, hipTotalDensity, hipTime, hipReadSlot, runTimeStepsHip
    ) where

import Control.Monad (forM, forM_, when)
import Data.Maybe (fromMaybe)
import qualified Data.Vector as V
import qualified Data.Vector.Storable as SV
import Data.Word (Word8)
import Foreign
import Foreign.C.String (CString, peekCString)
import Foreign.C.Types

import SolutionDomain

-- struct cfd_params (include/cfd_hip.h): 3 x int32, pad, 10 x double, 4 x int32 = 112 bytes
data Params = Params
    { pNx, pNy, pNz :: !Int32
    , pDx, pDy, pDz, pDt, pLambda, pSuth, pR, pCv, pK, pDyeF :: !Double
    , pDevice, pFlags, pGhostLo, pGhostHi :: !Int32 }

instance Storable Params where
    sizeOf _ = 112
    alignment _ = 8
    peek _ = error "Params: peek not needed"
    poke p (Params nx ny nz dx dy dz dt la su r cv k dy' dev fl gl gh) = do
        pokeByteOff p 0 nx >> pokeByteOff p 4 ny >> pokeByteOff p 8 nz
        forM_ (zip [16, 24 ..] [dx, dy, dz, dt, la, su, r, cv, k, dy']) $ \(o, v) -> pokeByteOff p o v
        pokeByteOff p 96 dev >> pokeByteOff p 100 fl >> pokeByteOff p 104 gl >> pokeByteOff p 108 gh

data CfdHandle
newtype HipGraph = HipGraph (Ptr CfdHandle)

foreign import ccall safe "cfd_create"
    c_create :: Ptr Params -> Ptr (Ptr Double) -> Ptr Word8 -> Ptr (Ptr CfdHandle) -> IO CInt
foreign import ccall safe "cfd_destroy" c_destroy :: Ptr CfdHandle -> IO ()
foreign import ccall safe "cfd_step" c_step :: Ptr CfdHandle -> CInt -> IO CInt
foreign import ccall safe "cfd_sweep" c_sweep :: Ptr CfdHandle -> CInt -> IO CInt
foreign import ccall safe "cfd_time_filter" c_filter :: Ptr CfdHandle -> IO CInt
foreign import ccall safe "cfd_sweep_and_filter" c_sweepAndFilter :: Ptr CfdHandle -> IO CInt
foreign import ccall safe "cfd_sweep_pair_and_filter" c_sweepPairAndFilter :: Ptr CfdHandle -> IO CInt
foreign import ccall safe "cfd_property_slice" c_propertySlice :: Ptr CfdHandle -> CInt -> CInt -> CInt -> Ptr Double -> IO CInt
foreign import ccall safe "cfd_sweep_begin" c_sweepBegin :: Ptr CfdHandle -> CInt -> CInt -> IO CInt
foreign import ccall safe "cfd_sweep_end" c_sweepEnd :: Ptr CfdHandle -> IO CInt
-- the split sweep with an edge of several planes, and the edge launched on another stream
foreign import ccall safe "cfd_sweep_begin_depth" c_sweepBeginDepth :: Ptr CfdHandle -> CInt -> CInt -> CInt -> IO CInt
foreign import ccall safe "cfd_sweep_end_on" c_sweepEndOn :: Ptr CfdHandle -> Ptr () -> IO CInt
-- slabs (several GPUs): halo ranges one or two planes deep, the neighbours' masks, availability of the one-pass tail
foreign import ccall safe "cfd_halo_planes_depth"
    c_haloPlanesDepth :: Ptr CfdHandle -> CInt -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> Ptr CSize -> IO CInt
foreign import ccall safe "cfd_set_ghost_active" c_setGhostActive :: Ptr CfdHandle -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "cfd_pair_available" c_pairAvailable :: Ptr CfdHandle -> Ptr CInt -> IO CInt
foreign import ccall safe "cfd_total_density" c_density :: Ptr CfdHandle -> Ptr Double -> IO CInt
foreign import ccall safe "cfd_read_fields" c_read :: Ptr CfdHandle -> CInt -> Ptr (Ptr Double) -> IO CInt
foreign import ccall safe "cfd_time" c_time :: Ptr CfdHandle -> Ptr CInt -> Ptr CInt -> IO CInt
foreign import ccall safe "cfd_last_error" c_last_error :: IO CString
-- checkpoint / restart: a slot written back (slot_rel counted from the current currModTime), then the counters
foreign import ccall safe "cfd_write_fields" c_write :: Ptr CfdHandle -> CInt -> Ptr (Ptr Double) -> IO CInt
foreign import ccall safe "cfd_set_time" c_setTime :: Ptr CfdHandle -> CInt -> CInt -> IO CInt

-- Several GPUs behind one handle: the whole grid goes in, the library cuts it into x-slabs (one per device, halo
-- planes and the density sum over RCCL), and the solve loop calls the cfd_multi_* twins of the functions above.
data CfdMulti
foreign import ccall safe "cfd_multi_create"
    c_multiCreate :: Ptr Params -> CInt -> Ptr CInt -> CInt -> Ptr (Ptr Double) -> Ptr Word8 -> Ptr (Ptr CfdMulti) -> IO CInt
foreign import ccall safe "cfd_multi_destroy" c_multiDestroy :: Ptr CfdMulti -> IO ()
foreign import ccall safe "cfd_multi_step" c_multiStep :: Ptr CfdMulti -> CInt -> IO CInt
foreign import ccall safe "cfd_multi_total_density" c_multiDensity :: Ptr CfdMulti -> Ptr Double -> IO CInt
foreign import ccall safe "cfd_multi_read_fields" c_multiRead :: Ptr CfdMulti -> CInt -> Ptr (Ptr Double) -> IO CInt
foreign import ccall safe "cfd_multi_time" c_multiTime :: Ptr CfdMulti -> Ptr CInt -> Ptr CInt -> IO CInt
-- checkpoint / restart of a decomposed run (the slots of all local slabs, then the counters)
foreign import ccall safe "cfd_multi_write_fields" c_multiWrite :: Ptr CfdMulti -> CInt -> Ptr (Ptr Double) -> IO CInt
foreign import ccall safe "cfd_multi_set_time" c_multiSetTime :: Ptr CfdMulti -> CInt -> CInt -> IO CInt
-- one process per GPU: rank 0 makes the 128-byte communicator id, every rank creates its slab with it
foreign import ccall safe "cfd_comm_unique_id" c_commUniqueId :: Ptr Word8 -> IO CInt
foreign import ccall safe "cfd_multi_create_rank"
    c_multiCreateRank :: Ptr Params -> CInt -> CInt -> Ptr Word8 -> Ptr (Ptr Double) -> Ptr Word8 -> Ptr (Ptr CfdMulti) -> IO CInt
-- hosts that bring their own communication library and want the whole step as one pass on their slab: masks two
-- planes deep once, then per step one three-plane exchange (cfd_halo_planes_depth with depth 3) and cfd_step_one_pass
foreign import ccall safe "cfd_set_ghost_active_depth"
    c_setGhostActiveDepth :: Ptr CfdHandle -> CInt -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "cfd_one_pass_step_available" c_onePassAvailable :: Ptr CfdHandle -> Ptr CInt -> IO CInt
foreign import ccall safe "cfd_step_one_pass" c_stepOnePass :: Ptr CfdHandle -> IO CInt

check :: String -> CInt -> IO ()
check what rc = when (rc /= 0) $ do
    msg <- c_last_error >>= peekCString
    ioError (userError (what ++ ": libcfdhip error " ++ show rc ++ ": " ++ msg))

-- | Flattens slot currModTime of an AdjGraph: seven property vectors in propIndex order, the
-- active mask, and the uniform geometry read from node 0 (every node carries the same copies,
-- SolutionDomain.hs:279-289).
marshalGraph :: AdjGraph -> (Params, [SV.Vector Double], SV.Vector Word8)
marshalGraph (AdjGraph (Grids g) cmt _) =
    let nodes = g V.! cmt
        n0 = nodes V.! 0
        len d = fromMaybe (error "edgeLen") (edgeLen n0 V.! d)
        prop i = SV.generate (V.length nodes) $ \j ->
            fromMaybe (error "property") (property (nodes V.! j) V.! i)
        act = SV.generate (V.length nodes) $ \j -> if active (nodes V.! j) then 1 else 0
        i32 = fromIntegral :: Int -> Int32
        params = Params (i32 (maxPos X + 1)) (i32 (maxPos Y + 1)) (i32 (maxPos Z + 1))
                        (len 0) (len 1) (len 2) timeStep sutherlandLambda sutherlandConstant
                        gasConstantR specificHeatCv 0.025 0.001 (-1) 0 0 0
    in (params, map prop [0 .. 6], act)

withHipGraph :: AdjGraph -> (HipGraph -> IO a) -> IO a
withHipGraph g body =
    let (params, fields, act) = marshalGraph g
    in with params $ \pp ->
       withMany SV.unsafeWith fields $ \fps ->
       withArray fps $ \fpp ->
       SV.unsafeWith act $ \ap ->
       alloca $ \hp -> do
           c_create pp fpp ap hp >>= check "cfd_create"
           h <- peek hp
           r <- body (HipGraph h)
           c_destroy h
           return r

-- | runSingleStep (Main.hs:192-200)
hipRunSingleStep :: HipGraph -> IO ()
hipRunSingleStep (HipGraph h) = c_step h 1 >>= check "cfd_step"

-- | totalDensity (Main.hs:269-270)
hipTotalDensity :: HipGraph -> IO Double
hipTotalDensity (HipGraph h) = alloca $ \p -> c_density h p >>= check "cfd_total_density" >> peek p

-- | (currModTime, currAbsTime)
hipTime :: HipGraph -> IO (Int, Int)
hipTime (HipGraph h) = alloca $ \a -> alloca $ \b -> do
    c_time h a b >>= check "cfd_time"
    (,) <$> (fromIntegral <$> peek a) <*> (fromIntegral <$> peek b)

-- | Seven property vectors of slot currModTime - slotRel.
hipReadSlot :: HipGraph -> Int -> Int -> IO [SV.Vector Double]
hipReadSlot (HipGraph h) n slotRel = do
    bufs <- forM [0 .. 6 :: Int] $ \_ -> mallocForeignPtrArray n
    withMany withForeignPtr bufs $ \ps -> withArray ps $ \pp ->
        c_read h (fromIntegral slotRel) pp >>= check "cfd_read_fields"
    return [SV.unsafeFromForeignPtr0 b n | b <- bufs]

-- | adjGraphToGrid (Main.hs:339-346) on the device: nx rows of ny values of Property prp at z == zLevel.
hipGraphToGrid :: HipGraph -> Property -> Int -> IO [[Double]]
hipGraphToGrid (HipGraph h) prp zLevel = do
    let nx = maxPos X + 1
        ny = maxPos Y + 1
    buf <- mallocForeignPtrArray (nx * ny)
    withForeignPtr buf $ \p ->
        c_propertySlice h (fromIntegral (fromEnum prp)) (fromIntegral zLevel) 0 p >>= check "cfd_property_slice"
    let v = SV.unsafeFromForeignPtr0 buf (nx * ny)
    return [SV.toList (SV.slice (x * ny) ny v) | x <- [0 .. nx - 1]]

-- | The solve loop of Main.hs:278-306 with the update on the GPU; prints the same five lines.
runTimeStepsHip :: AdjGraph -> Int -> IO ()
runTimeStepsHip g0 nsteps = withHipGraph g0 $ \hg ->
    forM_ [0 .. nsteps - 1] $ \step -> do
        hipRunSingleStep hg
        (cmt, cat) <- hipTime hg
        td <- hipTotalDensity hg
        putStrLn $ show (V.length (grids (allNodes g0) V.! 0))
        putStrLn $ "step: " ++ show step
        putStrLn $ "timeLevel: " ++ show cmt
        putStrLn $ "timeLevelAbsolute: " ++ show cat
        putStrLn $ "totalDensity: " ++ show td
        putStrLn " "
