/*
 * oracle/gen_balancer_script.c -- TEST INFRASTRUCTURE: golden bitrate-setpoint scripts
 * produced by the REFERENCE's own balancer (oracle/_ref/libceracoder_ref.so, compiled
 * unmodified from /root/reference/src/core).  Replays the three-phase scenario of
 * /root/reference/tests/test_integration.c:151-225 (min 500 / max 6000 kbit/s, SRT latency
 * 2000 ms, packet 1316 B; 10 x good network @500 ms, 10 x congestion @250 ms, 15 x recovery
 * @500 ms) and prints "<timestamp_ms> <new_bitrate_bps>" per control step.
 * usage: gen_balancer_script adaptive|aimd|fixed
 */
#include <stdio.h>
#include <string.h>

#include "balancer.h"
#include "balancer_runner.h"
#include "config.h"

int main(int argc, char **argv) {
    const char *algo = argc > 1 ? argv[1] : "adaptive";
    BelacoderConfig cfg;
    config_init_defaults(&cfg);
    cfg.min_bitrate = 500;
    cfg.max_bitrate = 6000;
    strncpy(cfg.balancer, algo, sizeof cfg.balancer - 1);
    BalancerRunner runner;
    if (balancer_runner_init(&runner, &cfg, NULL, 2000, 1316) != 0) return 1;
    BalancerInput in = {.buffer_size = 10, .rtt = 30.0, .send_rate_mbps = 5.0, .timestamp = 0, .pkt_loss_total = 0, .pkt_retrans_total = 0};
    for (int i = 0; i < 10; i++) { in.timestamp += 500; printf("%llu %d\n", (unsigned long long)in.timestamp, balancer_runner_step(&runner, &in).new_bitrate); }
    in.buffer_size = 150; in.rtt = 400.0;
    for (int i = 0; i < 10; i++) { in.timestamp += 250; printf("%llu %d\n", (unsigned long long)in.timestamp, balancer_runner_step(&runner, &in).new_bitrate); }
    in.buffer_size = 20; in.rtt = 50.0;
    for (int i = 0; i < 15; i++) { in.timestamp += 500; printf("%llu %d\n", (unsigned long long)in.timestamp, balancer_runner_step(&runner, &in).new_bitrate); }
    balancer_runner_cleanup(&runner);
    return 0;
}
