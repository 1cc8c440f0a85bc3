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
               ) where

import Control.Exception (bracket)
import Control.Monad (when)
import Data.Word (Word32)
import Foreign.C.String (CString, peekCString, withCString)
import Foreign.C.Types (CDouble (..), CInt (..))
import Foreign.Marshal.Alloc (alloca)
import Foreign.Ptr (Ptr, nullPtr)
import Foreign.Storable (peek)
import qualified Data.Vector.Storable.Mutable as VSM

import ShaderRendering (FragmentShader (..))
import Trace

data RmdfCtx
newtype HipRenderer = HipRenderer (Ptr RmdfCtx)

-- All calls block until the output buffer is complete and may take milliseconds: `safe`, so the
-- RTS can run other Haskell threads (FileModChecker's poller) meanwhile.
foreign import ccall safe "rmdf_create"
    c_rmdf_create :: Ptr (Ptr RmdfCtx) -> Ptr () -> IO CInt
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
    open = alloca $ \pctx -> do
        rc <- c_rmdf_create pctx nullPtr
        when (rc /= 0) $ do
            err <- rmdfLastError nullPtr
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
