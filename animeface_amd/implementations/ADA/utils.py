"""StyleGAN3 + adaptive discriminator augmentation on MI355X.

Mirrors the reference's ``implementations/ADA/utils.py`` (``train`` :14-92, ``main`` :94-191): the StyleGAN3 loop with the
``ADA`` pipe as ``augment`` and ``augment.update_p(real_prob)`` after the EMA update (:73).  The loop body itself is
``StyleGAN3.utils.TrainStep`` (it calls ``update_p`` whenever the augment object has one); the reference's
``MiniAccelerator`` (autocast + GradScaler wrapper) has no counterpart: activations are bf16, no loss scaling.
Under data parallelism ``ADA.update_p`` all-reduces its sign statistic so that ``p`` follows the global batch."""
import torch

from ...nnutils import get_device, sample_nnoise
from ... import distributed as dp
from ..StyleGAN3.utils import train, build_models, build_optimizers, SG3_ARGS  # noqa: F401
from .model import ADA

ADA_ARGS = {k: v for k, v in SG3_ARGS.items() if k != 'policy'}
ADA_ARGS.update(ada_interval=[4, 'update p every'], ada_target_kimg=[500, 'target k images'], ada_threshold=[0.6, 'ada threshold'])
_LOGFILE = ADA_ARGS.pop('logfile')
ADA_ARGS['logfile'] = _LOGFILE                                   # keep the reference's flag order: logfile last


def main(parser, dataset=None):
    from ...utils_argument import add_args
    parser = add_args(parser, ADA_ARGS)
    args = parser.parse_args()
    rank, world, _ = dp.init_distributed()
    device = get_device(not args.disable_gpu)
    amp = not args.disable_amp and not args.disable_gpu
    compute_dtype = torch.bfloat16 if amp else torch.float32
    const_input = sample_nnoise((args.num_test, args.latent_dim), device)
    G, G_ema, D = build_models(args, device, compute_dtype)
    dp.broadcast_module(G), dp.broadcast_module(G_ema), dp.broadcast_module(D)
    optimizer_G, optimizer_D = build_optimizers(G, D, args.lr, args.map_lr_scale, tuple(args.betas))
    reducer_G = dp.GradReducer(G.parameters()) if world > 1 else None
    reducer_D = dp.GradReducer(D.parameters()) if world > 1 else None
    augment = ADA(args.ada_interval, args.ada_target_kimg, args.ada_threshold, args.batch_size,
                  xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1,
                  brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1).to(device)
    if dataset is None:
        gen = torch.Generator(device='cpu').manual_seed(rank)
        batch = (torch.rand(args.batch_size, args.image_channels, args.image_size, args.image_size, generator=gen) * 2 - 1).to(device)
        dataset = [batch]
    if args.max_iters < 0:
        args.max_iters = len(dataset) * args.default_epochs
    return train(args.max_iters, dataset, args.latent_dim, const_input, G, G_ema, D, optimizer_G, optimizer_D,
                 args.gp_lambda, args.gp_every, augment, device, amp, args.save, args.logfile, log_every=args.log_every,
                 reducer_G=reducer_G, reducer_D=reducer_D)
