"""MIOpen find-results cache for the STOCK part of the network (ResNet18 + PSPNet: torch.nn convolutions on MIOpen).

With ``torch.backends.cudnn.benchmark = True`` MIOpen times every applicable solver on the GPU the first time a
process meets a convolution shape and remembers the winner in its *user find-db* (a text file under
``~/.config/miopen``).  On a fresh machine that search costs ~12 s for the inference batch of 8, ~50 s for 64 objects
and ~30 s for the training step -- before the first timed step.  ``morefusion_amd/miopen_db/`` holds the find-db MIOpen
wrote for exactly these shapes on an MI355X (gfx950, the MIOpen build named in the file names); ``enable()`` copies it
to a per-user scratch directory and points ``MIOPEN_USER_DB_PATH`` there, unless the caller already set one.  Shapes
that are not in it are searched and appended as usual; a different MIOpen build ignores the files (its own name
differs) and searches from scratch.  Configuration of the stock library, nothing of the hand-written path reads it."""
import os
import shutil
import tempfile

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def enable():
    """Call before the first convolution of the process (children inherit the variable).  Returns the path used."""
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return os.environ["MIOPEN_USER_DB_PATH"]
    if not os.path.isdir(_DB):
        return None
    dst = os.path.join(tempfile.gettempdir(), f"mf_miopen_db_{os.getuid()}")
    try:
        os.makedirs(dst, exist_ok=True)
        for name in os.listdir(_DB):
            target = os.path.join(dst, name)
            if not os.path.exists(target):  # (a db already grown by earlier processes is kept)
                tmp = f"{target}.{os.getpid()}.tmp"
                shutil.copyfile(os.path.join(_DB, name), tmp)
                os.replace(tmp, target)
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
