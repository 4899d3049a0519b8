# Example TRAINING config in the reference's format (configs/config_carpet_train.py of hbaatz/nerf-tex, block for block: TFRecord dataset,
# Proxy pixel / ray samplers in the carpet box, 4 images x 256 rays a batch, generated validation views, ParamNerf [1, 6], AlphaLoss(smape, mse),
# Adam 5e-4 with decay, 256 samples a ray) that needs nothing but this package: the dataset is a teacher network's renders,
#   python tools/make_example_dataset.py                      # -> datasets/example_carpet/{nerf, tfr}
#   python -m nerf_tex_amd.main configs/example_carpet_train.py
# A second call resumes from logs/example_carpet_train/checkpoints.
_emb = lambda n: {'module': 'network.model.FourierFeatures', 'n_freq_bands': n}
_box = {'module': 'network.proxy.AABB', 'b_0': [-1.5, -1.3, -.2], 'b_1': [1.3, 1.3, 1.9]}

config = {
    'module': 'network.train.Train',
    'target_path': 'logs/example_carpet_train',
    'override': True,
    'seed': 0,
    'train_dataset_config': {
        'module': 'network.dataset.Dataset',
        'data_loader_config': {'module': 'network.dataset.TFRecord', 'tfr_path': 'datasets/example_carpet/tfr'},
        'pixel_sampler_config': {'module': 'network.pixel_sampler.Proxy', 'n_samples': 256},
        'ray_sampler_config': {'module': 'network.ray_sampler.Proxy'},
        'proxy_config': dict(_box),
        'batchsize': 4,
        'shuffle_buffer_size': 100,
    },
    'val_dataset_config': {
        'module': 'network.dataset.Dataset',
        'data_loader_config': {
            'module': 'network.dataset.GenerateData', 'height': 128, 'width': 128, 'angle': 0.63, 'radius': 5.,
            'pose_dist_config': {'module': 'data.distribution.Constant', 'constants': [[.47, -.65, .6]]},
            'parameter_dist_config': {'module': 'data.distribution.Constant', 'constants': [[1, 1, 1, .1, 0, -.707, .707]]},
        },
        'pixel_sampler_config': {'module': 'network.pixel_sampler.Full'},
        'ray_sampler_config': {'module': 'network.ray_sampler.Proxy'},
        'proxy_config': dict(_box),
        'n_epochs': 1,
    },
    'model_config': {'module': 'network.model.ParamNerf', 'pos_embedding': _emb(10), 'dir_embedding': _emb(4), 'param_embedding': _emb(4), 'n_parameters': [1, 6]},
    'loss_config': {'module': 'network.loss.AlphaLoss', 'loss_fn': 'network.loss.smape', 'alpha_loss_fn': 'network.loss.mse'},
    'n_iters': 2000,
    'lrate': 5e-4,
    'lrate_decay': 500,
    'renderer_config': {'module': 'network.renderer.Renderer', 'n_samples': 256, 'perturb': True, 'render_chunk': 1024 * 32, 'net_chunk': 1024 * 64},
    'logger_config': {'module': 'network.logger.Logger', 'i_print': 100, 'i_img': 1000, 'i_checkpoint': 1000},
}
