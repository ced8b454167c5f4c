/*
 * sgw.h -- C ABI of the MI355X batched gridworld step/observation engine.
 *
 * Sorrel (social-ai-uoft/sorrel v1.4.0) is pure Python and has no FFI; the
 * interface a maintainer would bind is its class-based plugin API.  Each entry
 * point below names the reference interface it replaces for a batch of E
 * independent environments (paths relative to the reference root):
 *
 *   sgw_create      Environment.__init__ / setup_agents       sorrel/environment.py:36-54
 *                   + OneHotObservationSpec.__init__/generate_map
 *                                                             sorrel/observation/observation_spec.py:128-173
 *                   + Entity attribute table                  sorrel/entities/entity.py:29-39
 *                   + ActionSpec / MovingAgent.movement table sorrel/action/action_spec.py:23-26,
 *                                                             sorrel/agents/agent.py:187-213
 *   sgw_reset       Environment.reset -> Gridworld.create_world + populate_environment
 *                                                             sorrel/environment.py:72-79,
 *                                                             sorrel/worlds/gridworld.py:47-65,
 *                                                             sorrel/examples/treasurehunt/env.py:114-147
 *   sgw_observe     OneHotObservationSpec.observe -> visual_field -> shift
 *                                                             sorrel/observation/observation_spec.py:175-205,
 *                                                             sorrel/observation/visual_field.py:9-101,
 *                                                             sorrel/utils/helpers.py:48-77
 *   sgw_observe_full  ObservationSpec(full_view=True).observe  sorrel/observation/observation_spec.py:197-203,
 *                                                             sorrel/observation/visual_field.py:41-55
 *   sgw_step        Environment.take_turn                     sorrel/environment.py:81-93
 *                   -> Entity.transition sweep                sorrel/examples/treasurehunt/entities.py:69-85
 *                   -> Agent.transition (pov, act, reward)    sorrel/agents/agent.py:155-173
 *                   -> MovingAgent.act -> Gridworld.move      sorrel/agents/agent.py:215-225,
 *                                                             sorrel/worlds/gridworld.py:95-122
 *   sgw_observe_rows / sgw_act   Agent.transition of a policy-driven agent: pov, then act
 *                                                             sorrel/agents/agent.py:155-173, 215-225
 *   sgw_rollout     the turn loop of run_experiment           sorrel/environment.py:160-166
 *   sgw_reduce_metrics  world.total_reward read-out           sorrel/environment.py:193-199
 *
 * Conventions
 *   - every function returns 0 on success or a negative SGW_E* code and never
 *     throws; sgw_last_error() returns the message of the calling thread's
 *     last failure;
 *   - all array arguments are DEVICE pointers owned by the caller (PyTorch);
 *     the library owns only the engine handle, its constant tables and a small
 *     reduction workspace;
 *   - all kernels are enqueued asynchronously on the HIP stream passed as `stream`
 *     (a hipStream_t, NULL = default stream).  The calls that block the host are:
 *     sgw_create / sgw_destroy (allocate and upload the constant tables with blocking
 *     copies), sgw_get_status, sgw_get_step_time_ms and sgw_get_step_times_ms (they
 *     wait for the events / the status word they read), and -- only while timing is
 *     enabled with sgw_set_timing -- every 4096th sgw_step / sgw_observe, which waits
 *     for its oldest pair of events when the event pool wraps;
 *   - one engine per device, not thread-safe per engine (the reference is
 *     single-threaded);
 *   - there is no CPU fallback: without a HIP device every launch fails.
 *
 * Tensor layouts (C order, per-env contiguous)
 *   grid          uint8  [E][L][H][W]   entity TYPE id per cell (env e starts at e * grid_env_stride)
 *   agent_pos     uint8  [E][A][2]      (y, x) of each agent on `agent_layer`
 *   actions       uint8  [E][A]         action index per agent
 *   obs           float  [E][A][C][V][V], V = 2*vision_radius+1
 *   rewards       float  [E][A]
 *   total_reward  double [E]            world.total_reward (float64, agent order)
 */
#ifndef SGW_H
#define SGW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGW_MAX_TYPES 32
#define SGW_MAX_CHANNELS 16
#define SGW_MAX_CHOICES 8
#define SGW_MAX_ACTIONS 16
#define SGW_MAX_AGENTS 128 /* round 6 (64 before): up to 64 agents every kernel family applies (lane = agent on the wave- / workgroup-per-env kernels); 65..128 run on
                            * the ticket-ordered workgroup-per-env generic kernel, whatever the world's size (the reference steps any list: sorrel/environment.py:92-93) */
#define SGW_MAX_LAYERS 7   /* numpy sums <= 7 layers left to right; 8+ pairwise */
#define SGW_MAX_DIM 256    /* positions are uint8 */

/* entity transition rules (Entity.transition plugins the device can run) */
#define SGW_RULE_NONE 0
#define SGW_RULE_SPAWN 1   /* w.p. p replace own cell by one of n types, uniformly */
#define SGW_RULE_BECOME_IF 2 /* become rule_become if the cell of layer rule_layer in the same column holds a type in
                             * rule_mask; rule_layer < 0 = always (timers).  Cleanup: Pollution, Apple, beams
                             * (sorrel/examples/cleanup/entities.py:57-66,94-105, agents.py:190-205).  Columns are
                             * swept in ndenumerate order: a lower layer has already transitioned, a higher has not. */

#define SGW_NO_BORDER 255

/* observation post-processing (sgw_config.obs_post) */
/* action kinds (sgw_config.action_kind), SGW_AGENT_RULE_CLEANUP */
#define SGW_ACTION_MOVE 0
#define SGW_ACTION_CLEAN 1
#define SGW_ACTION_ZAP 2

#define SGW_OBS_POST_NONE 0            /* OneHotObservationSpec: the layer sum itself */
#define SGW_OBS_POST_CLIP255_DIV255 1  /* RGBObservationSpec: np.clip(obs, 0, 255) / 255 (observation_spec.py:483) */

/* counter-RNG stream ids: u32 = Philox4x32-10(ctr = {index>>2, turn, env, epoch<<4|stream},
 * key = {seed lo, seed hi})[index & 3] */
#define SGW_STREAM_SPAWN 0
#define SGW_STREAM_SPAWN_KIND 1
#define SGW_STREAM_ACTION 2
#define SGW_STREAM_PLACE 3
#define SGW_STREAM_DENSE 4
#define SGW_STREAM_DENSE_KIND 5
#define SGW_STREAM_TAG_INIT 6
#define SGW_STREAM_EXPLORE 7           /* index = agent: the epsilon test of SGW_ACT_QF32 (sgw_turn_epsilon) */

/* what Agent.act does (sgw_config.agent_rule) */
#define SGW_AGENT_RULE_MOVE 0 /* MovingAgent.act: reward = value of the target, then move (sorrel/agents/agent.py:215-225) */
#define SGW_AGENT_RULE_CLEANUP 2 /* CleanupAgent.act: move (turning the agent) or fire a clean / zap beam; reward summed over
                                  * all layers of the target cell (sorrel/examples/cleanup/agents.py:93-177);
                                  * needs sgw_bind_agent_dir */
#define SGW_AGENT_RULE_TAG 1  /* TagAgent.act: move, tag the first adjacent NotIt agent, reward for not being it
                               * (sorrel/examples/tag/agents.py:76-106); needs sgw_bind_agent_state */

/* sgw_step flags */
#define SGW_STEP_SWEEP 1u           /* run the entity-transition sweep first */
#define SGW_STEP_RANDOM_ACTIONS 2u  /* draw actions from STREAM_ACTION and STORE them to `actions` */
#define SGW_STEP_NO_OBS 4u          /* do not write observations (obs may be NULL) */
#define SGW_STEP_OBS_NEXT 8u        /* policy-driven (phased) stepping: do NOT write the observations of the stepped
                                     * agents; instead write the observation of agent `agent_end` (if < num_agents) from
                                     * the grid AFTER the moves of [agent_begin, agent_end) -- exactly what that agent's
                                     * pov() sees next (sorrel/agents/agent.py:167), so a policy turn costs 1 + A
                                     * launches instead of 1 + 2A */
#define SGW_STEP_OBS_NEXT_PACKED 16u /* with SGW_STEP_OBS_NEXT: `obs` is NOT the [E][A][C][V][V] tensor but one window per env,
                                     * [E][C][V][V] contiguous, which receives agent `agent_end`'s observation -- e.g. the
                                     * slot of that agent's replay buffer its pov() is about to be stored in
                                     * (sorrel/agents/agent.py:155-173: state = pov(); ...; add_memory(state, ...)), so the
                                     * observation is written once, where it will live */
#define SGW_STEP_NO_MOVE 32u         /* nobody acts: the entity sweep (with SGW_STEP_SWEEP) and the windows of [agent_begin, agent_end)
                                     * from the grid after it -- step 1 + 2 of a policy-driven turn in one launch when the windows
                                     * live in the [E][A][C][V][V] tensor (see sgw_act).  No rewards, no totals, no auto-reset;
                                     * SGW_STEP_RANDOM_ACTIONS / SGW_STEP_OBS_NEXT do not combine with it */
#define SGW_STEP_OBS_AGENT_MAJOR 64u /* `obs` is [A][E][C][V][V] (agent-major: agent a's windows of all envs are one contiguous [E][C*V*V] row, the shape a
                                     * replay ring row and a batched policy want) instead of [E][A][C][V][V].  Whole-turn calls (agent range 0 .. A, no
                                     * SGW_STEP_OBS_NEXT) of engines with SGW_CAP_OBS_AGENT_MAJOR -- the workgroup-per-env kernels (worlds above
                                     * 4 KiB): with SGW_STEP_NO_MOVE that is the sweep AND every agent's pre-move window into such rows in one launch */
#define SGW_STEP_DEFAULT (SGW_STEP_SWEEP)

/* error codes */
#define SGW_OK 0
#define SGW_EINVAL (-1)   /* rejected configuration / argument */
#define SGW_EHIP (-2)     /* HIP runtime error */
#define SGW_ENOMEM (-3)

/* bits of the device status word (sgw_get_status) */
#define SGW_STATUS_OOB_MOVE 1   /* an agent targeted a cell outside the grid (reference: IndexError / wrap) */
#define SGW_STATUS_BAD_ACTION 2 /* action index >= num_actions */
#define SGW_STATUS_BAD_TYPE 4   /* grid holds a type id >= num_types */
#define SGW_STATUS_BAD_POS 8    /* an agent position outside the grid was passed in (treated as (0, 0): memory-safe, result undefined) */

typedef struct sgw_config {
    int32_t height, width, layers;
    int32_t num_agents, vision_radius;
    int32_t num_types, num_channels, num_actions;
    int32_t agent_layer;   /* layer (z) every agent lives on */
    int32_t default_type;  /* Gridworld.default_entity: refills vacated cells */
    int32_t fill_type;     /* ObservationSpec.fill_entity_kind: out-of-bounds appearance */
    int32_t obs_post;      /* SGW_OBS_POST_*: applied to the float64 layer sum before the float32 cast */
    int8_t action_dy[SGW_MAX_ACTIONS]; /* in {-1,0,1}; (0,0) for non-move action names */
    int8_t action_dx[SGW_MAX_ACTIONS];
    uint8_t agent_type[SGW_MAX_AGENTS];
    double type_value[SGW_MAX_TYPES];     /* Entity.value */
    uint8_t type_passable[SGW_MAX_TYPES]; /* Entity.passable */
    uint8_t type_rule[SGW_MAX_TYPES];     /* SGW_RULE_* (Entity.has_transitions + transition) */
    double spawn_prob[SGW_MAX_TYPES];
    uint8_t spawn_count[SGW_MAX_TYPES];
    uint8_t spawn_choice[SGW_MAX_TYPES][SGW_MAX_CHOICES];
    double appearance[SGW_MAX_TYPES][SGW_MAX_CHANNELS]; /* ObservationSpec.entity_map[kind of type] */
    /* reset layout (populate_environment): per-layer fill and border types,
     * optional Bernoulli pre-seeding of the agent layer's interior, agents on
     * distinct interior cells */
    uint8_t layer_fill_type[8];
    uint8_t layer_border_type[8]; /* SGW_NO_BORDER = none */
    double dense_prob;
    uint8_t dense_count;
    uint8_t dense_choice[SGW_MAX_CHOICES];
    uint8_t agent_rule;     /* SGW_AGENT_RULE_*: what Agent.act does */
    uint8_t tag_it_type;    /* SGW_AGENT_RULE_TAG: type of an agent that is "it" (kind "It") */
    uint8_t tag_notit_type; /*                     ... that is not (kind "NotIt") */
    uint8_t reserved1[4];
    uint64_t seed;
    uint64_t first_env_id; /* global id of local env 0 (multi-GPU sharding) */
    int64_t num_envs;      /* E on this device */
    double tag_reward;     /* SGW_AGENT_RULE_TAG: TagAgent.reward_per_turn */
    /* SGW_RULE_BECOME_IF parameters, per type */
    int8_t rule_layer[SGW_MAX_TYPES];
    uint8_t rule_become[SGW_MAX_TYPES];
    uint32_t rule_mask[SGW_MAX_TYPES];
    /* SGW_AGENT_RULE_CLEANUP parameters */
    uint8_t action_kind[SGW_MAX_ACTIONS];
    int32_t beam_radius;
    uint8_t clean_beam_type; /* type of a freshly fired beam (its timer = a chain of SGW_RULE_BECOME_IF types) */
    uint8_t zap_beam_type;
    uint8_t reserved2[2];
    uint32_t beam_block_mask;    /* bit t: no beam is placed on a beam-layer cell holding type t (walls) */
    int32_t reward_total_factor; /* how many times act's reward enters total_reward (0 = 1; Cleanup = 2) */
    int64_t grid_env_stride; /* bytes between consecutive envs in `grid`; 0 = dense (L*H*W).  A stride that is a
                              * multiple of 16 lets worlds of any byte count use the 16-byte load/store kernels */
} sgw_config;

typedef struct sgw_engine sgw_engine;

int sgw_create(const sgw_config* cfg, sgw_engine** out);
void sgw_destroy(sgw_engine* eng);

int sgw_reset(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, double* total_reward,
              uint32_t epoch, void* stream);

/* Stateless observation of agents [agent_begin, agent_end) from the current
 * grid; obs rows of other agents are left untouched. */
int sgw_observe(sgw_engine* eng, const uint8_t* grid, const uint8_t* agent_pos, float* obs,
                int32_t agent_begin, int32_t agent_end, void* stream);

/* The whole map as an observation -- ObservationSpec(full_view=True).observe -> visual_field(location=None)
 * (sorrel/observation/observation_spec.py:140-142,197-203, sorrel/observation/visual_field.py:41-55): `out`
 * [E][C][H][W] (float, or uint8 with SGW_OBS_U8) receives every cell's appearance summed over the layers; no window, no
 * fill entity, the same for every agent. */
int sgw_observe_full(sgw_engine* eng, const uint8_t* grid, void* out, void* stream);

/* One take_turn for every env.  `turn` is Environment.turn AFTER its increment
 * (1 for the first turn of an epoch).  Agents [agent_begin, agent_end) are
 * stepped sequentially; pass (0, num_agents) for a whole turn.  A policy-driven
 * loop that must reproduce "agent i+1 observes agent i's move" calls
 * sgw_step once per agent with SGW_STEP_SWEEP only on the first call. */
int sgw_step(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* obs,
             float* rewards, double* total_reward, uint32_t epoch, uint32_t turn,
             int32_t agent_begin, int32_t agent_end, uint32_t flags, void* stream);

/* ---- Policy-driven turns without re-rendering (round 3) ----------------------------------------------------------
 * The reference's agent loop is pov -> get_action -> act, one agent after another (sorrel/agents/agent.py:155-173), and
 * agent j's pov shows the moves of agents < j.  A move changes at most two cells of the agent layer, so instead of
 * rendering a window per agent and launch:
 *   1. sgw_step(agent_begin = agent_end = 0, SGW_STEP_SWEEP | SGW_STEP_NO_OBS)      the entity sweep alone
 *   2. sgw_observe_rows(...)  (or sgw_observe into the [E][A][C][V][V] tensor)       EVERY agent's window, once
 *      (1 + 2 in ONE launch when the windows live in the tensor: sgw_step(0, A, SGW_STEP_SWEEP | SGW_STEP_NO_MOVE))
 *   3. per agent a, in order: policy(window a) -> actions[:, a];  sgw_act(a)         move agent a AND rewrite, in the
 *      windows of the agents after a that contain them, the <= 2 cells its move changed
 * gives every agent exactly the window the reference's pov() would build, at the cost of one fused turn plus A tiny
 * launches.  `rows[a]` (a HOST array of num_agents DEVICE pointers) is where agent a's window lives: element
 * rows[a][e * env_stride + (c * V + i) * V + j] for env e -- a row of that agent's replay buffer (env_stride = C*V*V) or
 * slot a of the observation tensor (rows[a] = obs + a*C*V*V, env_stride = A*C*V*V); element type = sgw_set_obs_format's.
 * sgw_observe_rows needs SGW_CAP_OBSERVE_ROWS (one-hot float32 windows of an instantiated layers / channels / radius),
 * sgw_act (SGW_CAP_ACT) serves every agent rule -- MovingAgent.act, TagAgent.act (the tagger's and its victim's cells are
 * repaired too; agent_state / state_at_pov kept as by sgw_step) and CleanupAgent.act (beam cells too; agent_dir kept) --
 * for any appearance table and float32 or uint8 windows; rows entries of agents
 * <= `agent` are ignored by sgw_act, NULL entries (or rows == NULL) are skipped.
 * sgw_act's optional extras keep the host out of the agent loop: `agent_action` (device, [E], element type
 * `action_kind`) is read INSTEAD of actions[:, agent] -- the policy's output tensor as it is, no narrowing copy; the
 * uint8 record actions[:, agent] is still written -- and `reward_row` (float [E]) / `action_row` (int64 [E]) receive a
 * second copy of the rewards / the actions: the rows of the agent's replay buffer (sorrel/buffers.py:46-63 stores
 * int64 actions and float32 rewards), so add_memory has nothing left to copy.  All three may be NULL.
 * sgw_act never runs the in-stream reset of sgw_set_auto_reset (that belongs to whole-turn sgw_step / sgw_rollout calls):
 * a policy-driven epoch loop resets with sgw_reset, as Environment.run_experiment does. */
#define SGW_CAP_OBSERVE_ROWS 1
#define SGW_CAP_ACT 2
#define SGW_CAP_RESOLVE 4      /* sgw_turn_resolve (speculative policy turns): plain movers, impassable agent types, float32 windows, at most 64 agents
                                * (any other rule set or agent count: sgw_verify_rows) */
#define SGW_CAP_OBS_AGENT_MAJOR 8   /* sgw_step accepts SGW_STEP_OBS_AGENT_MAJOR */
#define SGW_CAP_SWEEP_ROWS 16       /* sgw_sweep_observe_rows: steps 1 and 2 of the patched-window protocol in ONE launch (round 6: every kernel family where the
                                     * in-process specialiser is available; without hipRTC the headline's prebuilt instance only) */
#define SGW_ACT_U8 0
#define SGW_ACT_I32 1
#define SGW_ACT_I64 2
/* SGW_ACT_QF32: `agent_action` is the policy's VALUE output, float32 [E][num_actions] (contiguous): the act takes the first index
 * of the row's maximum itself (np.argmax / torch.argmax; a NaN counts as the maximum, as in both) -- the greedy half of the
 * reference's take_action (sorrel/models/pytorch/iqn.py:294-309: `random.random() > epsilon` -> argmax of the action values, else a
 * uniform action), one launch less per agent in a recorded turn.  The other half under the turn protocol (sgw_turn_act /
 * sgw_turn_act_rows): with probability epsilon[agent] (sgw_turn_epsilon; 0 until set) the action is the engine's own uniform draw
 * for (env, turn, agent) -- the action SGW_STEP_RANDOM_ACTIONS would take, stream SGW_STREAM_ACTION -- decided by
 * u32(SGW_STREAM_EXPLORE, index = agent) < floor(epsilon * 2^32).  sgw_act has no turn of its own: it reads epoch, turn in flight
 * (= completed + 1) and epsilon from the same device state, which its caller keeps current with sgw_turn_set (epsilon 0, the
 * default, needs neither).  The action taken is recorded in actions[:, agent] (uint8) and the replay row (int64) as for the
 * other kinds. */
#define SGW_ACT_QF32 3
int sgw_capabilities(sgw_engine* eng);
int sgw_observe_rows(sgw_engine* eng, const uint8_t* grid, const uint8_t* agent_pos, void* const* rows, int64_t env_stride,
                     int32_t agent_begin, int32_t agent_end, void* stream);
int sgw_act(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, void* const* rows, int64_t env_stride,
            float* rewards, double* total_reward, int32_t agent, const void* agent_action, int32_t action_kind,
            float* reward_row, int64_t* action_row, void* stream);
/* Steps 1 and 2 of that protocol in one launch (SGW_CAP_SWEEP_ROWS; float32 windows): the entity sweep of Environment.take_turn
 * (sorrel/environment.py:84-90; flags = SGW_STEP_SWEEP, or 0 for none) and then EVERY agent's window of the grid after the sweep
 * (Agent.pov, sorrel/agents/agent.py:158) into rows[a] + env * env_stride, env_stride == C * V * V + the bound row tail exactly; the tail
 * (sgw_bind_row_tail: what TagAgent.pov / CleanupObservation.observe append) is written behind every window.  Round 5: the wave-per-env
 * kernel's compile-time-shape instances of plain movers; round 6: every family -- step_big<..., ROWS>, the chunk-staging instances
 * (step_fast_rowsx), the generic kernel (step_kernel<..., ROWS>), Tag's whole-env instances -- as instances of their own.  Same
 * results as sgw_step(SGW_STEP_SWEEP | SGW_STEP_NO_OBS, agents [0, 0)) followed by sgw_observe_rows; the grid is read once and the
 * windows leave as the step kernel's one burst per env (config 3 at 65 536 envs: 183 us in two launches). */
int sgw_sweep_observe_rows(sgw_engine* eng, uint8_t* grid, const uint8_t* agent_pos, void* const* rows, int64_t env_stride,
                           uint32_t epoch, uint32_t turn, uint32_t flags, void* stream);

/* `num_turns` whole take_turns (all agents, in order) with one call -- Environment.run_experiment's inner loop
 * `while turn < max_turns: take_turn()` (sorrel/environment.py:160-166) for actions that need no host in between: drawn
 * on device (SGW_STEP_RANDOM_ACTIONS) or given up front.  Turn t of the call (t = 0 .. num_turns-1, Environment.turn =
 * first_turn + t) reads / writes its actions, observations and rewards `t * *_turn_stride` ELEMENTS after the pointers
 * passed (a stride of 0 makes every turn overwrite the same tensor; a ring of replay slots passes the slot size); grid,
 * agent_pos and total_reward hold the state after the last turn.  Where the step kernel supports it the turns run
 * inside ONE launch with the env's grid resident in LDS from turn to turn (no grid read / write-back and no kernel
 * boundary between turns); elsewhere the call is a loop of sgw_step launches.  Results are identical either way, and
 * identical to num_turns calls of sgw_step.  With sgw_set_auto_reset armed an epoch boundary inside the range is
 * honoured (the turn counter restarts at 1 in epoch + 1). */
int sgw_rollout(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* obs, float* rewards,
                double* total_reward, uint32_t epoch, uint32_t first_turn, uint32_t num_turns, int64_t obs_turn_stride,
                int64_t actions_turn_stride, int64_t rewards_turn_stride, uint32_t flags, void* stream);

/* out (device, 4 doubles) = { sum(total_reward), sum(total_reward^2), E, 0 },
 * summed in a fixed order (bitwise reproducible). */
int sgw_reduce_metrics(sgw_engine* eng, const double* total_reward, double* out, void* stream);

/* Per-env agent state: the CURRENT entity type of every agent, uint8 [E][A] (Tag: It / NotIt; it
 * survives resets, as TagAgent.it does).  Once bound, sgw_reset places the agents with these types
 * and sgw_step keeps them up to date; `state_at_pov` (uint8 [E][A], may be NULL) receives each
 * agent's type at the moment it observed (what TagAgent.pov appends to its observation).
 * sgw_init_agent_state fills `agent_state` with the configured agent types and, for
 * SGW_AGENT_RULE_TAG, draws the initial "it" agent of every env (sorrel/examples/tag/env.py:66-69). */
int sgw_bind_agent_state(sgw_engine* eng, uint8_t* agent_state, uint8_t* state_at_pov);
int sgw_init_agent_state(sgw_engine* eng, uint8_t* agent_state, void* stream);
/* Facing of every agent, uint8 [E][A]: 0 up, 1 right, 2 down, 3 left (MovingAgent.direction; CleanupAgent
 * starts at 2).  Caller-initialised; sgw_step updates it on move actions and reads it to aim beams. */
int sgw_bind_agent_dir(sgw_engine* eng, uint8_t* agent_dir);

/* Observation element type written by sgw_step / sgw_observe.  SGW_OBS_F32 (default) is the contract
 * format (the reference's replay buffer stores float32, sorrel/buffers.py:31).  SGW_OBS_U8 is a
 * compact extra for one-hot specs: the same [E][A][C][V][V] layout with uint8 counts (exactly the
 * float values, 4x fewer bytes); it is rejected for non one-hot appearance tables. */
#define SGW_OBS_F32 0
#define SGW_OBS_U8 1
int sgw_set_obs_format(sgw_engine* eng, int format);

/* Fill `actions` from STREAM_ACTION without stepping (what a RandomModel would choose). */
int sgw_random_actions(sgw_engine* eng, uint8_t* actions, uint32_t epoch, uint32_t turn, void* stream);

/* Synchronising: copies and clears the device status word (SGW_STATUS_* bits). */
int sgw_get_status(sgw_engine* eng, int32_t* status_out, void* stream);

/* Shape helpers (pure host arithmetic). */
int64_t sgw_obs_elems_per_env(const sgw_config* cfg);
int64_t sgw_grid_bytes_per_env(const sgw_config* cfg);
/* Algorithmic HBM bytes of one env-step (SURVEY.md 8d formula). */
int64_t sgw_algorithmic_bytes_per_env_step(const sgw_config* cfg);

/* Launch timing: with sgw_set_timing(eng, 1) every sgw_step / sgw_observe launch is bracketed by a pair of
 * HIP events on its stream.  sgw_get_step_time_ms returns (and clears) the sum and the number of launches since
 * the last read; sgw_get_step_times_ms copies the per-launch durations since the last read (oldest first, at most
 * `capacity`; *count = how many were written) to the HOST array `out_ms` and clears them; it returns 1 (not an error:
 * the durations written are valid) when launches were left out because `capacity` or the internal cap of 2^20 samples was
 * exceeded.  Both wait for the events they read. */
int sgw_set_timing(sgw_engine* eng, int enable);
int sgw_get_step_time_ms(sgw_engine* eng, double* total_ms, int64_t* launches);
int sgw_get_step_times_ms(sgw_engine* eng, float* out_ms, int64_t capacity, int64_t* count);

/* Auto-reset (Environment.run_experiment's epoch loop, sorrel/environment.py:148-171): once armed with
 * max_turns > 0, the sgw_step call that steps the LAST agent (agent_end == num_agents) of turn == max_turns also,
 * on the same stream and after the step kernel, copies total_reward to `episode_return` (device, double [E]; may be
 * NULL) and runs sgw_reset for epoch + 1 on the grid / agent_pos / total_reward it was given.  The caller goes on
 * with (epoch + 1, turn 1).  max_turns == 0 disarms. */
int sgw_set_auto_reset(sgw_engine* eng, uint32_t max_turns, double* episode_return);

/* Launch tuning of the wave-per-env step kernel.  Large float32 observation bursts of large batches run fastest
 * with fewer waves per CU than the register budget allows (fewer half-written observation streams open in HBM; see
 * DESIGN.md section 6); the only launch-time lever for that is the dynamic-LDS request, which bounds the workgroups
 * a CU admits.  wg_per_cu = 0 selects the documented automatic rule (a cap of 5 for whole-turn float32 observation
 * writes of >= 8 KiB per env when the batch's grids outgrow the caches or the emit is unstaged), 1..8 forces that
 * many workgroups per CU, -1 never caps.  sgw_launch_info writes a one-line description of what a whole-batch, whole-turn
 * sgw_step launches into buf: kernel variant, lanes per env, threads, `lds` = the dynamic-LDS bytes REQUESTED (the cap is
 * part of the request), `wg_per_cu` = the workgroups per CU the runtime admits for that request (occupancy query),
 * `cap` = policy and the cap in force (0 = none), `phase` = the kernel a policy-driven phase (one agent) takes. */
int sgw_set_wg_per_cu(sgw_engine* eng, int wg_per_cu);
int sgw_launch_info(sgw_engine* eng, char* buf, int64_t capacity);

/* ---- Options, the plan, specialised instances (round 4) -----------------------------------------------------------
 * The reference's plugin API takes ANY entity list, channel count and map size (sorrel/observation/observation_spec.py:128-173,
 * sorrel/entities/entity.py:9-68, sorrel/worlds/gridworld.py:36).  The step kernels are templates over exactly those
 * constants; sgw_create instantiates them for the engine's own world in-process (hipRTC; no compiler is spawned) -- EVERY
 * instance its plan counts on (whole turn, direct-store twin, rollout, walking variant, row kernels), so that a refusal of any of
 * them re-plans the whole engine for the prebuilt instances of the library (hipRTC absent, the library built without its embedded
 * sources, a compile error, a code object that does not load) instead of surfacing at a later call.  Code objects are kept on disk
 * (option "jit_cache_dir"; default <directory of libsgw.so>/jit_cache while that is the calling user's own directory, else
 * $XDG_CACHE_HOME/sgw_jit, $HOME/.cache/sgw_jit, /tmp/sgw_jit_cache_<uid>; mode 0700): a file is read only if it belongs to the
 * calling user, nobody else may write it, and its size and checksum are the ones its header states; the key names the compiler
 * (hipRTC version and library file), the architecture, the instance and the embedded source.
 *
 * sgw_set_option: every dispatcher knob is a (key, value) pair of strings; value NULL or "" restores the key's default, key
 * NULL restores all.  eng == NULL addresses the process-wide defaults that the NEXT sgw_create / sgw_plan copies; on a live
 * engine only the keys that do not shape its plan are accepted ("rows_mode", "act_lanes", "jit_verbose").  Keys (sorrel_amd/csrc/
 * options.h holds the table): jit, jit_cache, jit_cache_dir, jit_verbose, jit_own_rtc, jit_refuse (test hook), burst, pack3,
 * force_generic, fast_rules, rules_11k, fast_8k, force_big, group, phase_kernel, phase_rows, stage, stage_agents, fast_wg_per_cu,
 * big_threads, big_stage, big_wg_per_cu, big_walk, big_walk_blocks, big_walk_share, resolve_diag, rows_mode, act_lanes.  (Round 5
 * retired ten A/B hooks of closed experiments: static_radius, static_cleanup, rgb16, rules_8k, big_tag, stage_bytes, big_pad,
 * big_rot, big_walk_static, big_walk_stage -- an unknown key is SGW_EINVAL.)  No dispatcher knob is
 * an environment variable; the library reads SGW_DEBUG=1 (log lines of the specialiser on stderr), ROCM_PATH (which installation's
 * hipRTC: $ROCM_PATH/lib before /opt/rocm/lib) and, only to place the code-object cache, XDG_CACHE_HOME / HOME.
 *
 * sgw_plan: what sgw_create would decide for `cfg` on a device with `num_cus` compute units and `lds_per_workgroup` bytes of
 * LDS per workgroup -- kernel family, lanes per env, LDS layout, staging, the walk window, the instances to launch -- as pure
 * host arithmetic: no HIP call, no device needed (tests/test_plan.py enumerates it on the CPU).  It answers for a machine
 * where specialised instances are available unless option "jit" is 0. */
#define SGW_FAMILY_WAVE 1       /* step_fast: a wave per env (worlds <= 4 KiB; up to 11 KiB in large batches) */
#define SGW_FAMILY_WORKGROUP 2  /* step_big: a workgroup per env */
#define SGW_FAMILY_GENERIC 3    /* step_kernel: 16 / 32 lanes per env (small worlds of large batches), 64, or 256 (ticket-ordered) */
typedef struct sgw_plan_info {
    int32_t family;            /* SGW_FAMILY_* */
    int32_t lanes_per_env;     /* 16 / 32 / 64, or the workgroup size for a workgroup per env */
    int32_t threads;           /* per workgroup */
    int32_t grid_blocks;       /* workgroups of a whole-batch launch */
    int64_t lds_bytes;         /* dynamic LDS of the whole-turn kernel (before any occupancy cap) */
    int32_t env_lds;           /* ... of one env's slice */
    int32_t obs_stage;         /* bytes of window staging per wave (0: direct stores) */
    int32_t stage_agents;      /* agents per staged burst (0: the whole env at once, or no staging) */
    int32_t whole_env_burst;   /* the env's windows leave in ONE burst (compile-time shape) */
    int32_t big_stage;         /* step_big: bytes of window staging per wave */
    int32_t big_pitch;         /* step_big: LDS bytes between grid rows */
    int32_t onehot, rgb16, rules;
    int32_t specialised;       /* the plan counts on instances compiled for this engine */
    int32_t phase_kernel;      /* policy-driven phases of worlds above 4 KiB take the byte-gather phase kernel */
    int32_t rollout_in_one_launch;   /* sgw_rollout runs its turns inside one launch */
    int32_t walk_blocks;       /* step_big<..., WALK>: resident workgroups (0: not used) */
    int64_t walk_min_envs, walk_max_envs;   /* batches in (min, max] take the walking variant */
    int64_t big_stage_min_envs;             /* step_big stages its windows for batches above this */
    char kernel[192];              /* template-id of the whole-turn kernel (the specialised instance when `specialised`) */
    char kernel_prebuilt[192];     /* ... of the prebuilt instance used when the specialised one is unavailable ("-": none fits this plan) */
    char kernel_plain[192];        /* direct-store twin of a STAGE kernel (agent ranges, OBS_NEXT, unaligned tensors) */
    char kernel_rollout[192];      /* the instance with sgw_rollout's turn loop */
    char kernel_walk[192];
    char kernel_phase[96];         /* what a policy-driven phase (one agent) launches */
    char kernel_observe_rows[96];  /* sgw_observe_rows */
} sgw_plan_info;
int sgw_set_option(sgw_engine* eng, const char* key, const char* value);
int sgw_plan(const sgw_config* cfg, int32_t num_cus, int64_t lds_per_workgroup, sgw_plan_info* out);
/* ---- Row tails (round 4): what an agent's pov() appends to its flattened window, in-kernel ---------------------------------
 * TagAgent.pov appends whether the agent is "it" (sorrel/examples/tag/agents.py:57-65), CleanupObservation.observe the
 * positional code of the agent's cell (sorrel/examples/cleanup/agents.py:52-60, sorrel/observation/embedding.py:8-44).  With a
 * tail bound, sgw_observe_rows writes it right behind the window in every row (element C*V*V onwards; env_stride must be
 * >= C*V*V + tail_len) and sgw_act keeps it current (Tag: an agent tagged before its own pov reads 1), so the policy reads
 * the finished row and nothing is concatenated on the host.  SGW_TAIL_AGENT_IS_IT: one element, 1.0 where the agent's current
 * type is tag_it_type (needs sgw_bind_agent_state).  SGW_TAIL_POSITION_TABLE: tail_len elements table[(y*W + x)*tail_len ...]
 * of the agent's cell; `table` is a DEVICE pointer to float [H][W][tail_len] owned by the caller.  float32 rows only. */
#define SGW_TAIL_NONE 0
#define SGW_TAIL_AGENT_IS_IT 1
#define SGW_TAIL_POSITION_TABLE 2
int sgw_bind_row_tail(sgw_engine* eng, int kind, int tail_len, const float* table);

/* ---- A whole policy turn as ONE submission (round 4) ---------------------------------------------------------------
 * Agent.transition is pov -> get_action -> act, agent after agent (sorrel/agents/agent.py:155-173): 1 + A engine launches
 * with the policies' forward passes in between.  Three things change from turn to turn and used to arrive as kernel
 * arguments -- Environment.turn, the epoch, and the rows of the agents' replay rings (sorrel/buffers.py:46-63) -- so a turn
 * could not be recorded once and replayed.  They now live in device memory that the engine advances itself:
 *   sgw_turn_bind(rows)        (blocking, rare) the agents' replay rings: base pointers, capacity, the row the NEXT turn
 *                              fills, rows per turn (agents that share one Buffer advance it together); NULL: no replay rows.
 *                              Waits for everything submitted to the device so far (any stream), then writes the ring fields
 *                              only: epoch, turn and the exploration rates are never touched by it
 *   sgw_turn_set(epoch, turn)  (stream-ordered) the turns of the epoch completed so far: Environment.reset -> (epoch, 0)
 *   sgw_turn_begin(...)        sweep + EVERY agent's window into `obs` [E][A][C][V][V] for turn (completed + 1)
 *                              (= sgw_step(0, A, SGW_STEP_SWEEP | SGW_STEP_NO_MOVE) at the device's turn)
 *   sgw_turn_act(agent, ...)   = sgw_act with the windows in their slots of `obs` (the later agents' windows are repaired
 *                              there); reward and int64 action also go to the agent's ring row of the turn in flight
 *   sgw_turn_end(obs)          the windows of the turn -> each agent's ring row (what Agent.add_memory would copy), then
 *                              every ring advances and the turn counts as completed
 * Every call is asynchronous on `stream` and takes the same arguments every turn, so begin + A x (policy forward,
 * sgw_turn_act) + end can be captured in a hipGraph (torch.cuda.graph) and replayed without the host in the loop;
 * `obs` is the tensor the policies read -- a fixed address, which is what a recorded graph needs -- and the copy into the
 * ring rows is the price (one read + one write of the windows per turn).  sgw_turn_state reads the device's count back
 * (synchronising; tests and resynchronisation of the host's counters). */
typedef struct sgw_turn_rows {
    void* states[SGW_MAX_AGENTS];      /* device [capacity][E][row_elems], element type = sgw_set_obs_format's; NULL: windows not kept */
    float* rewards[SGW_MAX_AGENTS];    /* device [capacity][E]; may be NULL */
    int64_t* actions[SGW_MAX_AGENTS];  /* device [capacity][E]; may be NULL */
    float* dones[SGW_MAX_AGENTS];      /* device [capacity][E], zeroed for the turn's row; NULL when the ring's dones are all zero already */
    int64_t capacity[SGW_MAX_AGENTS];  /* rows of the ring; 0: this agent keeps no replay rows */
    int64_t row[SGW_MAX_AGENTS];       /* the row the next turn fills (Buffer.idx, + k for the k-th agent sharing the Buffer) */
    int64_t step[SGW_MAX_AGENTS];      /* rows the ring advances per turn (how many agents share the Buffer) */
    int64_t row_elems[SGW_MAX_AGENTS]; /* elements per env of a states row, >= C*V*V */
} sgw_turn_rows;
int sgw_turn_bind(sgw_engine* eng, const sgw_turn_rows* rows);
int sgw_turn_set(sgw_engine* eng, uint32_t epoch, uint32_t turn, void* stream);
int sgw_turn_begin(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* obs, float* rewards,
                   double* total_reward, uint32_t flags, void* stream);
int sgw_turn_act(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, void* obs, float* rewards,
                 double* total_reward, int32_t agent, const void* agent_action, int32_t action_kind, void* stream);
int sgw_turn_end(sgw_engine* eng, const void* obs, void* stream);
/* The same protocol with the windows in PER-AGENT rows (`rows[a]` + env * env_stride, as sgw_observe_rows / sgw_act take them; needs
 * SGW_CAP_OBSERVE_ROWS): sgw_turn_begin_rows = the sweep alone + every agent's window into its row AND -- by the device's row count --
 * into its replay row of the turn in flight; sgw_turn_act_rows repairs both copies; sgw_turn_end(eng, NULL, stream) then only
 * advances the rings.  No copy of the windows at the end of the turn (617 MB read + written at config 3's 65 536 envs), and each
 * policy reads a contiguous [E][C*V*V] row at a fixed address. */
int sgw_turn_begin_rows(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* rewards, double* total_reward,
                        void* const* rows, int64_t env_stride, uint32_t flags, void* stream);
int sgw_turn_act_rows(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, void* const* rows, int64_t env_stride,
                      float* rewards, double* total_reward, int32_t agent, const void* agent_action, int32_t action_kind, void* stream);
int sgw_turn_state(sgw_engine* eng, uint32_t* epoch_turn, int64_t* rows, void* stream);
/* ---- Speculative policy turns (round 5): many agents without A dependent (forward, act) pairs -----------------------------
 * Agent.transition runs agent after agent (sorrel/agents/agent.py:155-173): agent j's window shows the moves of the agents before
 * it.  A move changes two cells and a window is small, so for most (env, agent) pairs the action computed from the PRE-move window is
 * already the sequential one.  Protocol (needs SGW_CAP_RESOLVE):
 *   sgw_step(0, 0, SGW_STEP_SWEEP | SGW_STEP_NO_OBS)   the entity sweep alone
 *   sgw_observe_rows(rows, row_elems, 0, A)           every agent's pre-move window into `rows` [A][E][row_elems] float32
 *     (or sgw_turn_resolve(pass = 0))                  (rows[a] = rows + a * E * row_elems: agent-major, so agents that share a
 *                                                      model are ONE contiguous batch for its forward pass; pass 0 renders them for
 *                                                      ANY appearance table and window size, where there is no row kernel)
 *     (or, with SGW_CAP_OBS_AGENT_MAJOR, both in ONE launch: sgw_step(0, A, SGW_STEP_SWEEP | SGW_STEP_NO_MOVE | SGW_STEP_OBS_AGENT_MAJOR))
 *   fresh[A * E] <- policy(rows)                       one batched evaluation, int64 action indices in the rows' order
 *   sgw_turn_resolve(pass = 1, fresh, A * E)          writes the actions into actions[E][A], then resolves every env's moves in agent
 *             order WITHOUT touching the grid; renders, for each agent whose window an earlier mover touches, the window it really
 *             has when its turn comes; where that differs from its row the row is rewritten and the row's index (agent * E + env)
 *             appended to the pass's dirty list.  An env without a dirty agent has reached the fixed point: it is committed (movers'
 *             cells, agent_pos, rewards, the float64 total in agent order -- exactly what sgw_step with these actions would have
 *             written -- and, where given, reward / action once more in agent-major rows of a replay ring); later passes skip it.
 *   n <- counters[pass & 7]                            (the one thing the host reads back per pass)
 *   fresh[n] <- policy(rows[dirty_list[(pass & 1) * E * A + k]]), k < n;  sgw_turn_resolve(pass + 1, fresh, n);  until n == 0.
 * At most A passes (the first dirty agent of an env moves to a higher index every pass); measured three to four for 64 agents, with
 * a fifth of the rows re-evaluated in pass 2 and 0.3 % in pass 3 (profiles/r05_speculation_study.txt, r05_speculative_turn.txt).
 * With a policy that is a function of its window the result equals the sequential turn bit for bit.  Caller-owned scratch:
 * `scratch` 4 * E * A bytes (done flags, row states, the dirty bytes of the last pass at + 2 * E * A, previous moves), `dirty_list`
 * 2 * E * A int64, `counters` 8 uint32; their contents are initialised by pass 0 / 1.  (The list is appended with one atomic per env that has
 * dirty rows; from 8 192 envs on -- where that many atomics on one counter would serialise -- it is laid out by a scan over per-env counts,
 * two more small launches inside the call.)  `dirty_list` / `counters` may be NULL
 * (read the dirty bytes instead), and so may `new_actions` (the caller has written `actions` itself). */
int sgw_turn_resolve(sgw_engine* eng, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* rows, int64_t row_elems,
                     float* rewards, double* total_reward, uint8_t* scratch, int64_t* dirty_list, uint32_t* counters,
                     const int64_t* new_actions, int64_t n_new, float* reward_rows, int64_t* action_rows, int32_t pass, void* stream);
/* The resolve of a speculative turn for ANY agent rule (round 6; Tag, Cleanup, whatever rule comes next): play the current actions as ONE sequential turn on
 * a scratch copy of the state -- sgw_step with given actions, no sweep, observations on: its `obs` [E][A][C][V][V] are the windows the agents really have
 * when their turn comes (Agent.transition, sorrel/agents/agent.py:155-173), its state_at_pov Tag's "it" flags at pov time -- then sgw_verify_rows: row
 * (a, e) of `rows` [A][E][row_elems] (window + the bound row tail) that differs from that is rewritten from it and its index a * E + e appended to `list`
 * (*count = how many; zeroed by the call).  Evaluate the policy on those rows, sgw_apply_actions, play again -- until *count == 0: the scratch state then
 * IS the sequential turn's result (copy it over the state).  A positional tail (SGW_TAIL_POSITION_TABLE) is not compared: an agent's own cell does not
 * change before its own act.  Pays where the sequential loop is host-bound and the agents many (profiles/r06_speculation_study_rules.txt). */
int sgw_verify_rows(sgw_engine* eng, const float* obs, const uint8_t* state_at_pov, float* rows, int64_t row_elems, int64_t* list, uint32_t* count,
                    void* stream);
/* actions[e][a] = new_actions[k] for the k-th entry a * E + e of `list` (NULL: k itself), k < n; values outside [0, 255) become 255 (no ActionSpec has
 * it: SGW_STATUS_BAD_ACTION when played).  The dirty rows' fresh policy outputs into the [E][A] action tensor. */
int sgw_apply_actions(sgw_engine* eng, uint8_t* actions, const int64_t* list, const int64_t* new_actions, int64_t n, void* stream);
/* dst[k][:] = src[idx[k]][:] for k < n: rows of `row_elems` float32 (the dirty rows of a speculative pass as ONE contiguous batch for the
 * policy); src / dst 4-byte aligned device pointers, idx int64 on the device.  Asynchronous on `stream`; needs no engine. */
int sgw_gather_rows(const float* src, int64_t row_elems, const int64_t* idx, int64_t n, float* dst, void* stream);
/* Buffer.current_state (sorrel/buffers.py:143-154) for a recorded turn -- the frames a frame-stacking policy reads in front of its
 * window: the `count` rows of agent `agent`'s replay states BEFORE the row the turn in flight fills, oldest first, wrapping around
 * the ring, by the device's own row count -> out [count][E][row_elems] (device, element type = sgw_set_obs_format's).  Same
 * arguments every turn: recordable.  The ring must be bound with states (sgw_turn_bind); 1 <= count <= capacity. */
int sgw_turn_prev_rows(sgw_engine* eng, int32_t agent, int32_t count, void* out, void* stream);
/* The speculative turn's form of SGW_ACT_QF32 (a policy that returns action VALUES, sorrel/models/pytorch/iqn.py:294-309): for k < n,
 * out[k] = the action sgw_act(SGW_ACT_QF32) would take for agent a = row / E in env = row % E (row = idx[k]; idx NULL: row = k) from
 * values[k][0 .. num_actions): the first index of the maximum (NaN = maximum) or -- with probability epsilon[a] (sgw_turn_epsilon) -- the
 * engine's uniform draw for (env, `turn`, a) of `epoch`.  The draw is keyed, not consumed, so a window's action is a function of the
 * window and the fixed point of sgw_turn_resolve stays the sequential turn with exploration.  values float32 [n][num_actions], idx / out
 * int64 on the device; asynchronous on `stream`. */
int sgw_choose_actions(sgw_engine* eng, const float* values, const int64_t* idx, int64_t n, uint32_t epoch, uint32_t turn, int64_t* out,
                       void* stream);
/* Exploration rate of SGW_ACT_QF32 acts under the turn protocol: `agent` in [0, A) or -1 for every agent; epsilon in [0, 1].
 * Stream-ordered and kept in the device's turn state, so a recorded turn follows a decaying epsilon without being recorded again. */
int sgw_turn_epsilon(sgw_engine* eng, int32_t agent, double epsilon, void* stream);

/* out6 = { instances compiled, loaded from the disk cache, reused in memory, refused, ms spent compiling, ms spent loading }
 * of this process so far. */
int sgw_jit_stats(double* out6);
/* Compile (or find in the disk cache) the code object of one template instance -- e.g. "step_fast<true, 2, 5, 3, 32, 32>", the
 * `kernel` of an sgw_plan -- for `arch` ("gfx950") WITHOUT loading it: needs no device.  Fills a cache ahead of time (a build
 * machine, a container without a GPU) and lets tools look at the code (registers, scratch).  path_out receives the cache file:
 * "SGWJIT1\n", u32 name length, the lowered kernel name, the code object. */
int sgw_jit_compile(const char* instance, const char* arch, char* path_out, int64_t capacity);

const char* sgw_last_error(void);
const char* sgw_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SGW_H */
