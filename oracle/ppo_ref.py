"""PPO loss / GAE / batching of the oracle (test infrastructure, see oracle/__init__.py).

Follows /root/reference/molgym/ppo.py:18-95 (compute_loss, get_batch_generator, collect_data_batch),
/root/reference/molgym/buffer.py:74-82,104-110 (GAE-lambda, advantage standardisation) and
/root/reference/molgym/tools/util.py:61-87 (gradient norm, discount_cumsum).  These reference
functions import in the build container; tests/golden/*.npz pins this file against them.
"""
import numpy as np
import torch


def compute_loss_ref(ac, data, clip_ratio, vf_coef, entropy_coef, step_dtype=torch.float32):
    """float64 loss on float32 predictions (the buffer hands float64 numpy arrays to torch.as_tensor)."""
    pred = ac.step(data['obs'], data['act'], dtype=step_dtype)
    old_logp = torch.as_tensor(np.asarray(data['logp'], dtype=np.float64))
    adv = torch.as_tensor(np.asarray(data['adv'], dtype=np.float64))
    ret = torch.as_tensor(np.asarray(data['ret'], dtype=np.float64))
    return loss_from_pred(pred['logp'], pred['ent'], pred['v'], old_logp, adv, ret, clip_ratio, vf_coef,
                          entropy_coef)


def loss_from_pred(logp, ent, v, old_logp, adv, ret, clip_ratio, vf_coef, entropy_coef):
    ratio = torch.exp(logp - old_logp)
    obj = ratio * adv
    clipped_obj = ratio.clamp(1 - clip_ratio, 1 + clip_ratio) * adv
    policy_loss = -torch.min(obj, clipped_obj).mean()
    entropy_loss = -entropy_coef * ent.mean()
    vf_loss = vf_coef * (v - ret).pow(2).mean()
    loss = policy_loss + entropy_loss + vf_loss
    approx_kl = (old_logp - logp).mean()
    clipped = ratio.lt(1 - clip_ratio) | ratio.gt(1 + clip_ratio)
    clip_fraction = torch.as_tensor(clipped, dtype=torch.float32).mean()
    info = dict(policy_loss=policy_loss.item(), entropy_loss=entropy_loss.item(), vf_loss=vf_loss.item(),
                total_loss=loss.item(), approx_kl=approx_kl.item(), clip_fraction=clip_fraction.item())
    return loss, info


def gradient_norm_ref(parameters):
    """tools/util.py:61-69: 2-norm of the per-tensor gradient 2-norms (0.0 without any gradient)."""
    norms = [p.grad.detach().norm(2) for p in parameters if p.grad is not None]
    return torch.stack(norms).norm(2).item() if norms else 0.0


def discount_cumsum_ref(x, discount):
    out = np.zeros_like(np.asarray(x, dtype=np.float64))
    run = 0.0
    for t in range(len(x) - 1, -1, -1):
        run = x[t] + discount * run
        out[t] = run
    return out


def gae_ref(rews, vals, last_val, gamma, lam):
    """(adv, ret) of one trajectory."""
    r = np.append(np.asarray(rews, dtype=np.float64), last_val)
    v = np.append(np.asarray(vals, dtype=np.float64), last_val)
    deltas = r[:-1] + gamma * v[1:] - v[:-1]
    return discount_cumsum_ref(deltas, gamma * lam), discount_cumsum_ref(r, gamma)[:-1]


def normalize_adv_ref(adv):
    adv = np.asarray(adv, dtype=np.float64)
    return (adv - adv.mean()) / adv.std()


def batch_indices_ref(n, batch_size):
    """Mini-batch index arrays drawn from the GLOBAL numpy RNG exactly like ppo.get_batch_generator."""
    idx = np.random.permutation(np.arange(n))
    full = idx[:n // batch_size * batch_size].reshape(-1, batch_size)
    out = [b for b in full]
    rem = n % batch_size
    if rem:
        out.append(idx[-rem:])
    return out
