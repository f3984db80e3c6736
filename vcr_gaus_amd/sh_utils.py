"""SH <-> RGB helpers (`tools/sh_utils.py:114-117`).  The SH polynomial itself is evaluated by
the HIP preprocess kernel (csrc/preprocess.hip); no Python evaluation path is shipped."""
C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
