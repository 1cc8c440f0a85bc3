{-# LANGUAGE ForeignFunctionInterface #-}

-- | Haskell binding of librmdf.so (include/rmdf.h) for the rmdf viewer.
--
-- NOT compiled in this repository: the build image has no GHC.  It is the binding a
-- maintainer of blitzcode/ray-marching-distance-fields would add next to
-- ShaderRendering.hs; see INTEGRATION.md for the four-line change in App.draw.
--
-- `withHipRenderer` mirrors `withShaderRenderer :: FilePath -> FilePath -> (ShaderRenderer -> IO a) -> IO a`
-- (ShaderRendering.hs:60-110) minus the shader file (the kernels are ahead-of-time compiled for gfx950);
-- `drawHipTile` mirrors `drawShaderTile` (ShaderRendering.hs:151-196) but writes into the `MVector Word32`
-- that `FrameBuffer.fillFrameBuffer` hands out (FrameBuffer.hs:117-158) instead of issuing GL draws.

module RmdfFFI ( HipRenderer
               , withHipRenderer
               , drawHipTile
               , saveHipFrameBufferToPNG
               , rmdfLastError
                 -- * multi-GPU (one OS process per GPU; see INTEGRATION.md section 4)
               , HipCommId
               , hipCommGetUniqueId
               , withHipRendererOnRank
               , drawHipFrameSharded
               ) where

import Control.Exception (bracket)
import Control.Monad (void, when)
import qualified Data.ByteString as B
import qualified Data.ByteString.Unsafe as BU
import Data.Word (Word32, Word8)
import Foreign.C.String (CString, peekCString, withCString)
import Foreign.C.Types (CDouble (..), CInt (..), CSize (..))
import Foreign.Marshal.Alloc (alloca, allocaBytes)
import Foreign.Ptr (Ptr, castPtr, nullPtr)
import Foreign.Storable (peek)
import qualified Foreign.Storable
import qualified Data.Vector.Storable.Mutable as VSM

import ShaderRendering (FragmentShader (..))
import Trace

data RmdfCtx
newtype HipRenderer = HipRenderer (Ptr RmdfCtx)

-- All calls block until the output buffer is complete and may take milliseconds: `safe`, so the
-- RTS can run other Haskell threads (FileModChecker's poller) meanwhile.
foreign import ccall safe "rmdf_create"
    c_rmdf_create :: Ptr (Ptr RmdfCtx) -> Ptr () -> IO CInt
-- rmdf_create_ex hands the failure message back in the same call: with an unbound Haskell thread the runtime may run the
-- next foreign call on another OS thread (rmdf_last_error(NULL) is process-wide for that reason, but this is simpler)
foreign import ccall safe "rmdf_create_ex"
    c_rmdf_create_ex :: Ptr (Ptr RmdfCtx) -> Ptr CInt -> CString -> CSize -> IO CInt
foreign import ccall safe "rmdf_destroy"
    c_rmdf_destroy :: Ptr RmdfCtx -> IO ()
foreign import ccall unsafe "rmdf_last_error"
    c_rmdf_last_error :: Ptr RmdfCtx -> IO CString
foreign import ccall safe "rmdf_load_env_hdr"
    c_rmdf_load_env_hdr :: Ptr RmdfCtx -> CString -> IO CInt
foreign import ccall safe "rmdf_render_tile"
    c_rmdf_render_tile :: Ptr RmdfCtx -> CInt -> CInt -> CInt -> CInt -> CDouble -> CInt -> Ptr Word32 -> IO CInt

foreign import ccall safe "rmdf_save_png"
    c_rmdf_save_png :: CString -> Ptr Word32 -> CInt -> CInt -> IO CInt

rmdfLastError :: Ptr RmdfCtx -> IO String
rmdfLastError ctx = c_rmdf_last_error ctx >>= peekCString

-- | Open the renderer, load the lat/long environment (pre-convolved cache files are created next to
--   it on first use, exactly like buildPreConvolvedHDREnvMapCache), run the action, release everything.
withHipRenderer :: FilePath -> (HipRenderer -> IO a) -> IO a
withHipRenderer reflMapFn f =
    bracket open (\(HipRenderer ctx) -> c_rmdf_destroy ctx) f
  where
    open = alloca $ \pctx -> allocaBytes 1024 $ \errBuf -> do
        rc <- c_rmdf_create_ex pctx nullPtr errBuf 1024
        when (rc /= 0) $ do
            err <- peekCString errBuf
            traceAndThrow $ "withHipRenderer - Init failed:\n" ++ err
        ctx <- peek pctx
        rc' <- withCString reflMapFn $ c_rmdf_load_env_hdr ctx
        when (rc' /= 0) $ do
            err <- rmdfLastError ctx
            c_rmdf_destroy ctx
            traceAndThrow $ "withHipRenderer - Init failed:\n" ++ err
        return $ HipRenderer ctx

-- | drawShaderTile's signature plus the frame-buffer vector.  `Nothing` = whole frame, `Just idx` =
--   tile idx of the 8x8 grid; w/h/time are latched on the first tile of a frame by the library.
--   Returns `Left err` like loadAndCompileShaders does; the previous frame stays intact on failure.
drawHipTile :: HipRenderer -> FragmentShader -> Maybe Int -> Int -> Int -> Double
            -> VSM.IOVector Word32 -> IO (Either String ())
drawHipTile (HipRenderer ctx) shdEnum tileIdx w h time fbVec =
    VSM.unsafeWith fbVec $ \p -> do
        rc <- c_rmdf_render_tile ctx
                                 (fromIntegral $ fromEnum shdEnum)
                                 (maybe (-1) fromIntegral tileIdx)
                                 (fromIntegral w)
                                 (fromIntegral h)
                                 (realToFrac time)
                                 128 -- fragment.shd:634 MAX_STEPS
                                 p
        if rc == 0 then return $ Right ()
                   else Left <$> rmdfLastError ctx

-- | saveFrameBufferToPNG (FrameBuffer.hs:215-228) for a host that has the frame-buffer vector but no GL
--   texture to read back: rows flipped to top-down, alpha forced to 0xFF.  The viewer itself does not
--   need it -- its own screenshot path sees the texture fillFrameBuffer uploaded.
saveHipFrameBufferToPNG :: FilePath -> Int -> Int -> VSM.IOVector Word32 -> IO (Either String ())
saveHipFrameBufferToPNG fn w h fbVec =
    VSM.unsafeWith fbVec $ \p -> withCString fn $ \cfn -> do
        rc <- c_rmdf_save_png cfn p (fromIntegral w) (fromIntegral h)
        if rc == 0 then return $ Right ()
                   else Left <$> rmdfLastError nullPtr

-- ---------------------------------------------------------------------------------------------------------------------
-- Multi-GPU: the reference renders its 64 tiles one per frame (ShaderRendering.hs:49-52,183-193); here they are dealt to
-- the GPUs of a node, ONE RCCL gather per frame brings them to rank 0.  One OS process per GPU (start N viewers / workers;
-- rank 0 is the one with the window).  Rank 0 draws the unique id and ships its 128 bytes to the others by any channel.
-- ---------------------------------------------------------------------------------------------------------------------

type HipCommId = B.ByteString       -- RMDF_COMM_ID_BYTES = 128 bytes

foreign import ccall safe "rmdf_comm_get_unique_id"
    c_rmdf_comm_get_unique_id :: Ptr Word8 -> IO CInt
foreign import ccall safe "rmdf_comm_init"
    c_rmdf_comm_init :: Ptr RmdfCtx -> Ptr Word8 -> CInt -> CInt -> IO CInt
foreign import ccall safe "rmdf_comm_verify_deal"
    c_rmdf_comm_verify_deal :: Ptr RmdfCtx -> Ptr () -> IO CInt
foreign import ccall safe "rmdf_probe_tile_costs"
    c_rmdf_probe_tile_costs :: Ptr RmdfCtx -> CInt -> CInt -> CInt -> CDouble -> CInt -> Ptr Float -> IO CInt
foreign import ccall safe "rmdf_set_shard_costs"
    c_rmdf_set_shard_costs :: Ptr RmdfCtx -> Ptr Float -> IO CInt
foreign import ccall safe "rmdf_device_malloc"
    c_rmdf_device_malloc :: Ptr RmdfCtx -> CSize -> Ptr (Ptr ()) -> IO CInt
foreign import ccall safe "rmdf_device_free"
    c_rmdf_device_free :: Ptr RmdfCtx -> Ptr () -> IO CInt
foreign import ccall safe "rmdf_render_frame_sharded_device"
    c_rmdf_render_frame_sharded_device :: Ptr RmdfCtx -> CInt -> CInt -> CInt -> CDouble -> CInt
                                       -> Ptr () -> Ptr () -> Ptr () -> Ptr () -> IO CInt
foreign import ccall safe "rmdf_copy_to_host"
    c_rmdf_copy_to_host :: Ptr RmdfCtx -> Ptr () -> Ptr () -> CSize -> Ptr () -> IO CInt

hipCommGetUniqueId :: IO (Either String HipCommId)
hipCommGetUniqueId = allocaBytes 128 $ \p -> do
    rc <- c_rmdf_comm_get_unique_id p
    if rc == 0 then Right <$> B.packCStringLen (castPtr p, 128)
               else Left <$> rmdfLastError nullPtr

-- | withHipRenderer for rank `rank` of `nranks` (GPU ordinal = rank): environment, cost-aware deal of the 64 tiles for the
--   view that will be drawn (shader `shd`, w, h, `time`, `maxSteps` -- every rank computes the same deal by itself from a probe
--   frame of THAT view), RCCL communicator, device buffers (released with the renderer).
--   The action gets the renderer and a per-frame draw function (rank 0 receives the whole frame in its vector, the other
--   ranks pass a dummy vector and get nothing back).
withHipRendererOnRank :: FilePath -> HipCommId -> Int -> Int -> FragmentShader -> Int -> Int -> Double -> Int
                      -> (HipRenderer -> (FragmentShader -> Double -> VSM.IOVector Word32 -> IO (Either String ())) -> IO a) -> IO a
withHipRendererOnRank reflMapFn commId rank nranks shd0 w h time0 maxSteps f =
    withRenderer $ \hr@(HipRenderer ctx) -> do
        let slots = (64 + nranks - 1) `div` nranks
            tile  = (w `div` 8) * (h `div` 8) * 4
            check what rc = when (rc /= 0) $ rmdfLastError ctx >>= \e -> traceAndThrow (what ++ ": " ++ e)
        allocaBytes (64 * 4) $ \cost -> do
            c_rmdf_probe_tile_costs ctx (fromIntegral $ fromEnum shd0) (fromIntegral w) (fromIntegral h) (realToFrac time0)
                                    (fromIntegral maxSteps) cost
                >>= check "rmdf_probe_tile_costs"
            c_rmdf_set_shard_costs ctx cost >>= check "rmdf_set_shard_costs"
        BU.unsafeUseAsCString commId $ \p ->
            c_rmdf_comm_init ctx (castPtr p) (fromIntegral rank) (fromIntegral nranks) >>= check "rmdf_comm_init"
        -- collective: do all ranks hold the deal this rank computed from its own probe?  The verdict is the same on every rank; on a
        -- mismatch all of them drop to the static deal (a function of the rank count alone)
        agreed <- c_rmdf_comm_verify_deal ctx nullPtr
        when (agreed /= 0) $ c_rmdf_set_shard_costs ctx nullPtr >>= check "rmdf_set_shard_costs"
        let dmalloc n = alloca $ \pp -> c_rmdf_device_malloc ctx (fromIntegral n) pp >>= check "rmdf_device_malloc" >> peek pp
            dfree p   = when (p /= nullPtr) $ void $ c_rmdf_device_free ctx p
        -- rmdf_device_malloc is a bare device allocation: rmdf_destroy does not own it, so the bracket releases it
        bracket (do g  <- if rank == 0 then dmalloc (nranks * slots * tile) else return nullPtr
                    fr <- if rank == 0 then dmalloc (w * h * 4) else return nullPtr
                    sh <- if rank == 0 then return nullPtr else dmalloc (slots * tile)
                    return (g, fr, sh))
                (\(g, fr, sh) -> dfree g >> dfree fr >> dfree sh) $ \(dGathered, dFrame, dOwn) -> do
            let dShard = if rank == 0 then dGathered else dOwn                -- the root renders into its own slot
            f hr $ \shd time fbVec -> do
                rc <- c_rmdf_render_frame_sharded_device ctx (fromIntegral $ fromEnum shd) (fromIntegral w) (fromIntegral h)
                                                         (realToFrac time) (fromIntegral maxSteps) dShard dGathered dFrame nullPtr
                rc' <- if rc == 0 && rank == 0
                           then VSM.unsafeWith fbVec $ \p -> c_rmdf_copy_to_host ctx (castPtr p) dFrame (fromIntegral $ w * h * 4) nullPtr
                           else return rc
                if rc' == 0 then return $ Right () else Left <$> rmdfLastError ctx
  where
    -- rmdf_config { device = rank, reserved = 0 }: eight C ints
    withRenderer g = allocaBytes 32 $ \cfg -> do
        mapM_ (\i -> pokeElemOff' cfg i (if i == 0 then fromIntegral rank else 0)) [0 .. 7 :: Int]
        bracket (open cfg) (\(HipRenderer ctx) -> c_rmdf_destroy ctx) g
    pokeElemOff' :: Ptr CInt -> Int -> CInt -> IO ()
    pokeElemOff' = Foreign.Storable.pokeElemOff
    open cfg = alloca $ \pctx -> allocaBytes 1024 $ \errBuf -> do
        rc <- c_rmdf_create_ex pctx cfg errBuf 1024
        when (rc /= 0) $ peekCString errBuf >>= \e -> traceAndThrow ("withHipRendererOnRank - Init failed:\n" ++ e)
        ctx <- peek pctx
        rc' <- withCString reflMapFn $ c_rmdf_load_env_hdr ctx
        when (rc' /= 0) $ rmdfLastError ctx >>= \e -> c_rmdf_destroy ctx >> traceAndThrow ("withHipRendererOnRank - Init failed:\n" ++ e)
        return $ HipRenderer ctx

-- | One frame of the sharded path; a synonym kept for symmetry with drawHipTile.
drawHipFrameSharded :: (FragmentShader -> Double -> VSM.IOVector Word32 -> IO (Either String ()))
                    -> FragmentShader -> Double -> VSM.IOVector Word32 -> IO (Either String ())
drawHipFrameSharded draw = draw
