"""Second, independent restatement of the reference's hot-path arithmetic in numpy.  TEST INFRASTRUCTURE ONLY.

Written from the reference sources (cited per function), not from oracle/nerf_oracle.c: the two
restatements check each other in tests/test_oracle.py.  Ray generation and sampling are stepped in
float32 exactly as the C expressions evaluate (numpy rounds every float32 operation, no fused
multiply-add), so those are compared bit for bit; everything that goes through expf/sinf/cosf is
evaluated in float64 and compared within a few float32 ulps.

PARITY UNPINNED: the reference ships no test vectors and cannot be built here.
"""
import numpy as np

f32 = np.float32
DELTA = 0.01   # nerf.c:35 -- constant sample spacing inside the opacity
T_STOP = 0.01  # nerf.c:43 -- forward early termination threshold


def generate_ray(position, rotation, focal, width, height, u, v):
    """data.c:271-289, float32 step by step."""
    rot = np.asarray(rotation, f32).reshape(9)
    x = (f32(u) - f32(width) * f32(0.5)) / f32(focal)
    y = -(f32(v) - f32(height) * f32(0.5)) / f32(focal)
    z = f32(-1.0)
    norm = np.sqrt(x * x + y * y + z * z, dtype=f32)
    x, y, z = x / norm, y / norm, z / norm
    d = np.array([rot[0] * x + rot[1] * y + rot[2] * z,
                  rot[3] * x + rot[4] * y + rot[5] * z,
                  rot[6] * x + rot[7] * y + rot[8] * z], f32)
    return np.asarray(position, f32).copy(), d


def sample_points(ray_o, ray_d, S, near, far):
    """nerf.c:186-198: t = near + (far - near) * s / (S - 1), point = o + t d; rows [pos | dir]."""
    near, far = f32(near), f32(far)
    X = np.empty((S, 6), f32)
    for s in range(S):
        t = near + (far - near) * f32(s) / f32(S - 1)
        X[s, 0:3] = ray_o + t * ray_d
        X[s, 3:6] = ray_d
    return X


def positional_encoding(x, L):
    """nerf.c:129-145: [x, then per frequency l and component i: sin(2^l pi x_i), cos(2^l pi x_i)];
    the argument is formed in double (M_PI is a double) and rounded to float for sinf/cosf."""
    x = np.asarray(x, f32)
    out = [x.astype(np.float64)]
    for l in range(L):
        freq = np.float64(f32(2.0) ** f32(l))
        arg = (freq * np.pi * x.astype(np.float64)).astype(f32).astype(np.float64)
        out.append(np.stack([np.sin(arg), np.cos(arg)], axis=-1).reshape(-1))
    return np.concatenate(out)


def encode_points(X, Lp, Ld):
    """nerf.c:147-162: position block then direction block per row."""
    return np.stack([np.concatenate([positional_encoding(r[:3], Lp), positional_encoding(r[3:], Ld)]) for r in X])


def activation(y):
    """nerf.c:7-19: density = relu(y0); colour = sigmoid(y1..3)."""
    y = np.asarray(y, np.float64)
    return np.maximum(0.0, y[:, 0]), 1.0 / (1.0 + np.exp(-y[:, 1:4]))


def composite_forward(dens, rgb, target, R, S):
    """nerf.c:21-61: pixel, error and summed squared error; the loop stops after the sample that
    takes the transmittance below T_STOP.  Also returns how close each ray came to the threshold."""
    dens = np.asarray(dens, np.float64).reshape(R, S)
    rgb = np.asarray(rgb, np.float64).reshape(R, S, 3)
    pixel = np.zeros((R, 3))
    margin = np.full(R, np.inf)
    for r in range(R):
        T = 1.0
        for s in range(S):
            alpha = 1.0 - np.exp(-dens[r, s] * DELTA)
            pixel[r] += alpha * T * rgb[r, s]
            T *= 1.0 - alpha
            margin[r] = min(margin[r], abs(T - T_STOP) / T_STOP)
            if T < T_STOP:
                break
    err = pixel - np.asarray(target, np.float64).reshape(R, 3)
    return pixel, err, float((err * err).sum()), margin


def composite_gradient_literal(dens, rgb, err, R, S):
    """nerf.c:63-127 loop for loop (O(S^3) per ray: small cases only).  No early stop here."""
    dens = np.asarray(dens, np.float64).reshape(R, S)
    rgb = np.asarray(rgb, np.float64).reshape(R, S, 3)
    err = np.asarray(err, np.float64).reshape(R, 3)
    out = np.zeros((R, S, 4))
    for r in range(R):
        alphas = 1.0 - np.exp(-dens[r] * DELTA)
        trans = np.ones(S)
        for s in range(1, S):
            trans[s] = trans[s - 1] * (1.0 - alphas[s - 1])
        weights = alphas * trans
        for s in range(S):
            out[r, s, 1:4] = weights[s] * err[r] * rgb[r, s] * (1.0 - rgb[r, s])
            dalpha = DELTA * np.exp(-dens[r, s] * DELTA)
            g = dalpha * trans[s] * float(err[r] @ rgb[r, s])
            for t in range(s + 1, S):
                dtrans = -dalpha
                for k in range(s + 1, t + 1):
                    dtrans *= 1.0 - alphas[k - 1]
                g += alphas[t] * dtrans * float(err[r] @ rgb[r, t])
            out[r, s, 0] = g if dens[r, s] > 0.0 else 0.0
    return out.reshape(R * S, 4)


def composite_gradient_closed(dens, rgb, err, R, S):
    """The same sums in closed form.  The inner product of nerf.c:112-114 runs over k = s+1..t and
    multiplies (1 - alpha_{k-1}), i.e. alpha_s .. alpha_{t-1}: it equals T_t / T_s, so
    G_s = dalpha_s * (T_s q_s - sum_{t>s} alpha_t q_t T_t / T_s), q_t = err . colour_t."""
    dens = np.asarray(dens, np.float64).reshape(R, S)
    rgb = np.asarray(rgb, np.float64).reshape(R, S, 3)
    err = np.asarray(err, np.float64).reshape(R, 3)
    alphas = 1.0 - np.exp(-dens * DELTA)
    om = 1.0 - alphas
    out = np.zeros((R, S, 4))
    for r in range(R):
        q = rgb[r] @ err[r]
        trans = np.concatenate([[1.0], np.cumprod(om[r])[:-1]])
        out[r, :, 1:4] = (alphas[r] * trans)[:, None] * err[r][None, :] * rgb[r] * (1.0 - rgb[r])
        for s in range(S):
            dalpha = DELTA * np.exp(-dens[r, s] * DELTA)
            tail = 0.0
            for t in range(s + 1, S):
                tail += alphas[r, t] * q[t] * np.prod(om[r, s:t])
            g = dalpha * (trans[s] * q[s] - tail)
            out[r, s, 0] = g if dens[r, s] > 0.0 else 0.0
    return out.reshape(R * S, 4)


LAST_DELTA = 1e10  # include/nerf_spec.h NERF_LAST_DELTA


def depths(S, near, far, offsets=None):
    """float32 depths exactly as sample_points evaluates them (offsets in units of one spacing)"""
    near, far = f32(near), f32(far)
    out = np.empty(S, f32)
    for s in range(S):
        fs = f32(s) if offsets is None else f32(s) + f32(offsets[s])
        out[s] = near + (far - near) * fs / f32(S - 1)
    return out


def spacing_render(sigma, colour, t):
    """Opt-in quadrature (NOT the reference's): delta_s = t_{s+1} - t_s, last = LAST_DELTA, no early
    stop.  One ray, float64.  Returns the pixel and the per-sample weights."""
    delta = np.append(np.diff(np.asarray(t, np.float64)), LAST_DELTA)
    alpha = 1.0 - np.exp(-np.asarray(sigma, np.float64) * delta)
    T = np.concatenate([[1.0], np.cumprod(1.0 - alpha)[:-1]])
    w = T * alpha
    return w @ np.asarray(colour, np.float64), w, alpha, T, delta


def spacing_gradient(sigma, colour, err, t):
    """analytic gradient of 1/2 |pixel - target|^2 w.r.t. (density, colour pre-sigmoid) per sample:
    dL/dsigma_s = delta_s ((1 - alpha_s) T_s q_s - sum_{t>s} w_t q_t); relu' applied by the caller's sigma > 0"""
    _, w, alpha, T, delta = spacing_render(sigma, colour, t)
    colour = np.asarray(colour, np.float64)
    q = colour @ np.asarray(err, np.float64)
    wq = w * q
    tail = np.concatenate([np.cumsum(wq[::-1])[::-1][1:], [0.0]])
    g = np.zeros((len(w), 4))
    g[:, 0] = np.where(np.asarray(sigma) > 0, delta * ((1.0 - alpha) * T * q - tail), 0.0)
    g[:, 1:] = w[:, None] * np.asarray(err, np.float64)[None, :] * colour * (1.0 - colour)
    return g


def swish(x):
    return x / (1.0 + np.exp(-x))


def dswish(x):
    sg = 1.0 / (1.0 + np.exp(-x))
    return sg * (1.0 + x * (1.0 - sg))


def mlp_forward(W1, W2, X):
    """two bias-free layers with swish between (include/nerf_spec.h; the upstream mlp submodule is absent)."""
    pre = np.asarray(X, np.float64) @ np.asarray(W1, np.float64).T
    act = swish(pre)
    return pre, act, act @ np.asarray(W2, np.float64).T


def mlp_backward(W1, W2, X, pre, act, err_out):
    """summed (not averaged) weight gradients, hidden error and input gradient"""
    W1, W2, X, err_out = (np.asarray(a, np.float64) for a in (W1, W2, X, err_out))
    dW2 = err_out.T @ act
    err_hidden = (err_out @ W2) * dswish(pre)
    dW1 = err_hidden.T @ X
    return dW1, dW2, err_hidden, err_hidden @ W1


def adamw(w, g_sum, m, v, t, lr, batch, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """one AdamW step on gradient g_sum / batch; t is the step count after the increment"""
    w, g, m, v = (np.asarray(a, np.float64) for a in (w, g_sum, m, v))
    g = g / batch
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    alpha_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    return w * (1 - lr * wd) - alpha_t * m / (np.sqrt(v) + eps), m, v
