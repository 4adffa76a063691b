from pokerrl_amd.rl.neural.TorchPolicyAgent import PolicyNet, TorchPolicyAgent  # noqa: F401
