"""
Run configuration object. Field names follow the reference (PokerRL/rl/base_cls/TrainingProfileBase.py:16-139); only the
fields the tabular hot path reads are kept: n_seats, eval_modes_of_algo, eval_stack_sizes,
module_args["env" | "lbr"], game_cls_str, env_builder_cls_str, DEBUGGING, HAVE_GPU, DISTRIBUTED, CLUSTER, name,
device_inference. The additional `module_args["tree_backend"]` field is ignored: there is one backend, the MI355X one.
"""
import copy
import os


class TrainingProfileBase:
    def __init__(self, name, log_verbose, log_export_freq, checkpoint_freq, eval_agent_export_freq, game_cls, env_bldr_cls,
                 start_chips, eval_modes_of_algo, eval_stack_sizes, module_args, path_data=None, local_crayon_server_docker_address="localhost",
                 cluster_address=None, DEBUGGING=False, redis_head_adr=None, device_inference="cpu", DISTRIBUTED=False, CLUSTER=False):
        self.name = name
        self.log_verbose = log_verbose
        self.log_export_freq = log_export_freq
        self.checkpoint_freq = checkpoint_freq
        self.eval_agent_export_freq = eval_agent_export_freq
        self.module_args = module_args
        self.game_cls_str = game_cls.__name__
        self.env_builder_cls_str = env_bldr_cls.__name__
        self.n_seats = module_args["env"].n_seats
        assert self.n_seats == 2, "the MI355X hot path is heads-up (like the reference's CFR / BR / LBR)"
        if start_chips is None:
            self.start_chips = game_cls.DEFAULT_STACK_SIZE
        else:
            self.start_chips = int(start_chips)
        self.eval_modes_of_algo = eval_modes_of_algo
        if eval_stack_sizes is None:
            self.eval_stack_sizes = [[self.start_chips for _ in range(self.n_seats)]]
        else:
            self.eval_stack_sizes = copy.deepcopy(eval_stack_sizes)
        self.DEBUGGING = DEBUGGING
        self.DISTRIBUTED = DISTRIBUTED or CLUSTER
        self.CLUSTER = CLUSTER
        self.device_inference = device_inference
        try:
            import torch
            self.HAVE_GPU = torch.cuda.is_available()
        except Exception:
            self.HAVE_GPU = False
        self.local_crayon_server_docker_address = local_crayon_server_docker_address
        self.redis_head_adr = redis_head_adr if redis_head_adr is not None else cluster_address
        self.path_data = path_data if path_data is not None else os.path.join(os.path.expanduser("~"), "poker_ai_data")
        self.path_checkpoint = os.path.join(self.path_data, "checkpoint")
        self.path_agent_export_storage = os.path.join(self.path_data, "eval_agent")
        self.path_log_storage = os.path.join(self.path_data, "logs")
