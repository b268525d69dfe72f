"""Saving results (reference t2v_pipeline.py:166-189 uses torchvision.io.write_video / ToPILImage; neither
torchvision nor PyAV exists in the ROCm image).  write_video tries torchvision, then PyAV, and otherwise writes the
frames losslessly as an animated PNG next to the requested path (same base name, .png) so nothing is silently lost."""
import os

import torch


def to_pil_images(images):
    """(B,3,H,W) uint8 -> list of PIL images."""
    from PIL import Image
    return [Image.fromarray(img.permute(1, 2, 0).contiguous().numpy()) for img in images]


def write_video(path, frames, fps=24, crf="5"):
    """frames (T,H,W,3) uint8 (CPU).  Returns the path actually written."""
    frames = frames.to(torch.uint8).contiguous()
    try:
        import torchvision
        torchvision.io.write_video(path, frames.numpy(), fps=fps, options={"crf": str(crf)})
        return path
    except Exception:
        pass
    try:
        import av
        with av.open(path, mode="w") as container:
            stream = container.add_stream("libx264", rate=fps)
            stream.height, stream.width = frames.shape[1], frames.shape[2]
            stream.options = {"crf": str(crf)}
            for f in frames.numpy():
                for packet in stream.encode(av.VideoFrame.from_ndarray(f, format="rgb24")):
                    container.mux(packet)
            for packet in stream.encode():
                container.mux(packet)
        return path
    except Exception:
        pass
    from PIL import Image
    out = os.path.splitext(path)[0] + ".png"
    imgs = [Image.fromarray(f) for f in frames.numpy()]
    imgs[0].save(out, save_all=True, append_images=imgs[1:], duration=int(round(1000 / fps)), loop=0)
    return out
