"""Independent numpy restatement of the path (float32 arrays, BLAS matmul, np.tanh) used to
cross-check the C oracle within tolerance — written from src/perform_step.jl:3-47 and
SURVEY.md §3.5, sharing no code with oracle/lrnde_oracle.c."""
import numpy as np

f32 = np.float32
C = [0.161, 0.327, 0.9, 0.9800255409045097]
A = {2: [0.161],
     3: [-0.008480655492356989, 0.335480655492357],
     4: [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
     5: [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
     6: [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
     7: [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]}
BT = [-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
      0.5823571654525552, -0.45808210592918697, 0.015151515151515152]


def gelu(x):
    x = x.astype(np.float64)
    return (0.5 * x * (1 + np.tanh(np.sqrt(2 / np.pi) * (x + 0.044715 * x ** 3)))).astype(f32)


class NpMlp:
    def __init__(self, D, H, p, time_dep=True, act="tanh"):
        td = int(time_dep)
        self.D, self.H, self.td, self.act = D, H, td, act
        o = 0
        self.W1 = p[o:o + H * (D + td)].reshape(D + td, H).T.copy(); o += H * (D + td)   # (H, D+td)
        self.b1 = p[o:o + H].copy(); o += H
        self.W2 = p[o:o + D * (H + td)].reshape(H + td, D).T.copy(); o += D * (H + td)   # (D, H+td)
        self.b2 = p[o:o + D].copy()

    def __call__(self, u, t):  # u: (B, D)
        B = u.shape[0]
        tcol = np.full((B, 1), t, dtype=f32)
        x = np.concatenate([u, tcol], axis=1) if self.td else u
        h = x @ self.W1.T + self.b1
        h = np.tanh(h) if self.act == "tanh" else (gelu(h) if self.act == "gelu" else h)
        h = np.concatenate([h.astype(f32), tcol], axis=1) if self.td else h.astype(f32)
        return (h @ self.W2.T + self.b2).astype(f32)


def rms(x):
    return f32(np.sqrt(np.mean(np.square(x.astype(np.float64)))))


def tsit5_step(f, uprev, k1, t, dt, abstol, reltol):
    t, dt = f32(t), f32(dt)
    ks = [k1]
    cs = [C[0], C[1], C[2], C[3], 1.0, 1.0]
    xs = {}
    for s in range(2, 8):
        acc = sum(f32(a) * k for a, k in zip(A[s], ks))
        x = (uprev + dt * acc).astype(f32)
        xs[s] = x
        ks.append(f(x, f32(t + f32(cs[s - 2]) * dt)))
    u, g6 = xs[7], xs[6]
    utilde = dt * sum(f32(b) * k for b, k in zip(BT, ks))
    resid = utilde / (f32(abstol) + np.maximum(np.abs(uprev), np.abs(u)) * f32(reltol))
    eest = rms(resid)
    den = rms(u - g6)
    stiff = f32(0) if den == 0 else f32(abs(rms(ks[6] - ks[5]) / (den + np.finfo(f32).eps)) / f32(3.5068))
    return dict(u=u, k7=ks[6], eest=eest, reg_error=f32(eest * dt), reg_stiff=stiff)
