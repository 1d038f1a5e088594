"""GAN training CLI and step loop (reference gan/gan_train_for_shadow.py:28-318), same flag names.

    python -m hypelcnn_amd.gan.gan_train_for_shadow --loader_name SyntheticDataLoader --path gulfport \
        --gan_type cycle_gan --batch_size 2048 --step 1000 --pairing_method dummy

One loop iteration = one `session.run(global_step_inc_op)` of the reference with its sequential RunTrainOpsHooks:
every phase (generator, discriminator[, feature discriminator]) is a pre-planned list of HIP launches replayed as
a HIP graph; the paired samples stay resident in HBM."""
import argparse
import json
import os
from types import SimpleNamespace

import numpy
import torch

from hypelcnn_amd.common.cmd_parser import add_parse_cmds_for_json_loader, add_parse_cmds_for_loaders, \
    add_parse_cmds_for_loggers, add_parse_cmds_for_opt, add_parse_cmds_for_trainers, type_ensure_strtobool
from hypelcnn_amd.backend import Ref
from hypelcnn_amd.common.common_nn_ops import get_loader_from_name
from hypelcnn_amd.common.common_ops import replace_abbrs
from hypelcnn_amd.gan.wrapper_registry import get_sampling_map, get_wrapper_dict
from hypelcnn_amd.gan.wrappers import gan_common as C


def add_parse_cmds_for_app(parser):
    b = type_ensure_strtobool
    parser.add_argument("--gan_type", nargs="?", type=str, default="cycle_gan")
    parser.add_argument("--use_identity_loss", nargs="?", type=b, default=True)
    parser.add_argument("--identity_loss_weight", nargs="?", type=float, default=0.5)
    parser.add_argument("--regularization_support_rate", nargs="?", type=float, default=0.0)
    parser.add_argument("--cycle_consistency_loss_weight", nargs="?", type=float, default=10.0)
    parser.add_argument("--nce_loss_weight", nargs="?", type=float, default=10.0)
    parser.add_argument("--tau", nargs="?", type=float, default=0.07)
    parser.add_argument("--patches", nargs="?", type=int, default=6)
    parser.add_argument("--embedded_feat_size", nargs="?", type=int, default=2)
    parser.add_argument("--validation_steps", nargs="?", type=int, default=1000)
    parser.add_argument("--validation_sample_count", nargs="?", type=int, default=300)
    parser.add_argument("--generator_lr", nargs="?", type=float, default=0.0002)
    parser.add_argument("--discriminator_lr", nargs="?", type=float, default=0.0001)
    parser.add_argument("--gen_discriminator_lr", nargs="?", type=float, default=0.0001)
    parser.add_argument("--discriminator_reg_scale", nargs="?", type=float, default=0.00001)
    parser.add_argument("--gen_disc_reg_scale", nargs="?", type=float, default=0.0001)
    parser.add_argument("--pairing_method", nargs="?", type=str, default="random")
    parser.add_argument("--master", nargs="?", type=str, default="")      # TF1 parameter-server flags: accepted,
    parser.add_argument("--ps_tasks", nargs="?", type=int, default=0)     # ignored (vestigial)
    parser.add_argument("--task", nargs="?", type=int, default=0)


def read_hsi_data(loader, data_set, shadow_map, pairing_method, sampling_method_map):
    """reference gan_common.py:385-392"""
    normal, shadow = sampling_method_map[pairing_method].get_sample_pairs(data_set, loader, shadow_map)
    return normal[..., :data_set.get_casi_band_count()], shadow[..., :data_set.get_casi_band_count()]


class PairIterator:
    """load_op (reference :147-168): paired (normal, shadow) spectra resident on the device, shuffle_and_repeat over
    `epoch` epochs, per-sample regulariser swap (perform_shadow_augmentation_random :171-182), batch(drop_remainder)."""

    def __init__(self, normal, shadow, batch_size, iteration_count, shadow_ratio, reg_support_rate, device, seed=1234,
                 backend=None):
        n = normal.shape[0]
        self.backend = backend
        self._out = None
        self.normal = torch.as_tensor(normal.reshape(n, -1), dtype=torch.float32).to(device)
        self.shadow = torch.as_tensor(shadow.reshape(n, -1), dtype=torch.float32).to(device)
        self.batch_size = batch_size
        self.epochs = max(1, (iteration_count * batch_size) // n)
        self.ratio = None if shadow_ratio is None else torch.as_tensor(shadow_ratio, dtype=torch.float32).to(device)
        self.rate = reg_support_rate
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed)
        order = [torch.randperm(n, generator=self.gen) for _ in range(self.epochs)]
        self.order = torch.cat(order).to(device)
        self.pos = 0

    def next_batch(self):
        # data parallel: the ranks share the seeded order, draw the global batch (batch_size per rank x world) and keep
        # the pairs rank::world of it
        import torch.distributed as dist
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() \
            else (1, 0)
        take = self.batch_size * world
        if self.pos + take > self.order.numel():
            return None
        idx = self.order[self.pos:self.pos + take][rank::world].contiguous()
        self.pos += take
        be = self.backend
        if be is None:  # no kernel library handed in (host-side tests of the iterator): the same arithmetic in torch
            x, y = self.normal.index_select(0, idx), self.shadow.index_select(0, idx)
            if self.rate > 0 and self.ratio is not None:
                u1 = (torch.rand(take, generator=self.gen) * 0.98 + 0.01)[rank::world].to(x.device).unsqueeze(1)
                u2 = (torch.rand(take, generator=self.gen) * 0.98 + 0.01)[rank::world].to(x.device).unsqueeze(1)
                x = torch.where(u1 < self.rate, y * self.ratio, x)
                y = torch.where(u2 < self.rate, x / self.ratio, y)
            return x.contiguous(), y.contiguous()
        # one library launch: gather + regulariser swap (hypel_gather_pairs_f32), into buffers that live as long as
        # the iterator; the two uniform draws per pair come from the host generator (seeded, identical on every rank)
        nb, bands = int(idx.numel()), self.normal.shape[1]
        if self._out is None or self._out[0].numel() != nb * bands:
            self._out = (torch.empty(nb * bands, dtype=torch.float32, device=self.normal.device),
                         torch.empty(nb * bands, dtype=torch.float32, device=self.normal.device))
        u1 = u2 = ratio = None
        if self.rate > 0 and self.ratio is not None:
            u = torch.rand(2, take, generator=self.gen) * 0.98 + 0.01
            u = u[:, rank::world].contiguous().reshape(-1).to(self.normal.device, non_blocking=True)
            u1, u2, ratio = Ref(u, 0), Ref(u, nb), Ref(self.ratio.reshape(-1))
            self._keep = u
        be.call("gather_pairs_f32", Ref(self.normal.reshape(-1)), Ref(self.shadow.reshape(-1)), Ref(idx), nb, bands, ratio,
                u1, u2, float(self.rate), Ref(self._out[0]), Ref(self._out[1]))
        self._keep_idx = idx
        return self._out[0].view(nb, bands), self._out[1].view(nb, bands)


def create_stats(generated_y, images_x, shadow_ratio):
    """create_stats_tensor (reference gan_common.py:315-330): band-ratio statistics of generated / input * ratio and
    the JS-divergence style scalars the reference tracks as its only quality signal."""
    ratio = (generated_y / images_x * shadow_ratio).double()
    finite = torch.isfinite(ratio).all(dim=1)
    ratio = ratio[finite]
    mean, std = ratio.mean(0), ratio.std(0, unbiased=False)

    def kl(p, q):
        return torch.where(p != 0, p * torch.log(p / q), torch.zeros_like(p)).sum()

    def js(p, q):
        m = 0.5 * (p + q)
        return 0.5 * kl(p, m) + 0.5 * kl(q, m)

    zeros = torch.zeros_like(mean)
    return float(js((mean - 1).abs(), zeros).abs()), float(js((mean + std - 1).abs(), zeros).abs()), mean, std


def get_log_suffix(flags):
    patch_size = flags.neighborhood * 2 + 1
    suffix = f"{flags.loader_name.lower():s}_{flags.gan_type.lower():s}_{patch_size:d}x{patch_size:d}_" \
             f"regsup{flags.regularization_support_rate:.2f}_batch{flags.batch_size:d}".replace(".", "")
    if flags.use_identity_loss is True:
        suffix += f"_idnty{flags.use_identity_loss:.2f}".replace(".", "")
    return replace_abbrs(suffix, {"dataloader": "ldr"})


def save_gan_checkpoint(sess, log_dir, step):
    os.makedirs(log_dir, exist_ok=True)
    path = os.path.join(log_dir, f"model.ckpt-{int(step)}.npz")
    from hypelcnn_amd.classify.monitored_session_runner import is_chief
    if is_chief():  # data parallel: parameters are identical on every rank, rank 0 writes
        numpy.savez(path, **{k.replace("/", "|"): v for k, v in sess.state_dict().items()})
    return path


def gan_train(train_ops, iterator, log_dir, max_steps, save_checkpoint_steps=None, log_every=1000):
    """reference :80-144: loop until the step budget or the data runs out; returns the last global step."""
    sess = train_ops.ctx.session()
    while sess.global_step < max_steps:
        batch = iterator.next_batch()
        if batch is None:
            break
        train_ops.run_step(*batch)
        step = sess.global_step
        if log_every and step % log_every == 0:
            print(f"Starting train step: {step}", {k: round(v, 5) for k, v in train_ops.losses().items()})
        if save_checkpoint_steps and log_dir and step % save_checkpoint_steps == 0:
            save_gan_checkpoint(sess, log_dir, step)
    return sess.global_step


def run_session(params, base_log_path, backend=None):
    flags = SimpleNamespace(**params)
    print("Args:", json.dumps(vars(flags), indent=3))
    log_dir = f"{base_log_path}_{get_log_suffix(flags)}"
    neighborhood = 0  # 1x1 spectral "patches" (reference :249)
    loader = get_loader_from_name(flags.loader_name, flags.path)
    data_set = loader.load_data(neighborhood, True)
    shadow_map, shadow_ratio = loader.load_shadow_map(neighborhood, data_set)
    normal, shadow = read_hsi_data(loader, data_set, shadow_map, flags.pairing_method, get_sampling_map())
    bands = data_set.get_casi_band_count()

    wrapper = get_wrapper_dict(flags)[flags.gan_type]
    wrapper.backend = backend
    tower, images_x, images_y = C.new_gan_tower(bands)
    the_gan_model = wrapper.define_model(images_x, images_y)
    the_gan_loss = wrapper.define_loss(the_gan_model)
    train_ops = wrapper.define_train_ops(the_gan_model, the_gan_loss, max_number_of_steps=flags.step,
                                         generator_lr=flags.generator_lr, discriminator_lr=flags.discriminator_lr,
                                         gen_discriminator_lr=flags.gen_discriminator_lr)
    sess = train_ops.ctx.session()
    iterator = PairIterator(normal, shadow, flags.batch_size, flags.step, shadow_ratio,
                            flags.regularization_support_rate, sess.backend.device, backend=sess.backend)
    gan_train(train_ops, iterator, log_dir, flags.step, save_checkpoint_steps=flags.validation_steps)
    save_gan_checkpoint(sess, log_dir, sess.global_step)
    # final quality statistic on a sample of the pairs (the reference's PeerValidationHook tracks the same scalars)
    gen = sess.compile_phase(tower, min(flags.validation_sample_count, normal.shape[0]),
                             outputs=the_gan_loss.generate_outputs[:1], key="validate")
    xs = iterator.normal[:gen.plan.nb]
    src = iterator.shadow[:gen.plan.nb] if getattr(wrapper, "_swap_inputs", False) else xs
    b = gen.plan.buffers
    gen.set_input("y" if "in:y" in b and "in:x" not in b else "x", src)
    gen.forward()
    ratio = shadow_ratio if shadow_ratio is not None else numpy.ones(bands, numpy.float32)
    div_mean, div_upper, _, _ = create_stats(gen.value(the_gan_loss.generate_outputs[0], copy=False), src,
                                             torch.as_tensor(ratio, dtype=torch.float32).to(src.device))
    return [div_upper, div_mean]


def main(argv=None):
    parser = argparse.ArgumentParser()
    for add in (add_parse_cmds_for_loaders, add_parse_cmds_for_loggers, add_parse_cmds_for_trainers,
                add_parse_cmds_for_json_loader, add_parse_cmds_for_app, add_parse_cmds_for_opt):
        add(parser)
    flags, _ = parser.parse_known_args(argv)
    params = dict(vars(flags))
    if flags.flag_config_file:
        params.update(json.load(open(flags.flag_config_file, "r")))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    print("Output divergence values:", run_session(params, flags.base_log_path))


if __name__ == "__main__":
    main()
